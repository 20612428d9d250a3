#!/bin/bash
# same-box A/B of an environment switch: tools/ab3.sh <ENVVAR> <workload> [more workloads]
V=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for on in 0 1; do
      if [ $on = 1 ]; then export $V=1; else unset $V; fi
      python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$V=' + '$on', 'ms/step', d['ms_per_step'])"
    done
  done
done
