#!/bin/bash
# same-box A/B of library builds over whole workloads: tools/ab4.sh "<lib> <lib> ..." <workload> [workloads]   ("cur" = in-tree build)
LIBS=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for L in $LIBS; do
      if [ "$L" = "cur" ]; then unset MSSEG_LIB; else export MSSEG_LIB=$PWD/$L; fi
      python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$L', 'ms/step', d['ms_per_step'])"
    done
  done
done
