#!/bin/bash
# A/B of library builds on one box: tools/ab.sh <lib1> <lib2> ...  (paths relative to the repo root; "cur" = the in-tree build)
for L in "$@"; do
  if [ "$L" = "cur" ]; then unset MSSEG_LIB; else export MSSEG_LIB=$PWD/$L; fi
  echo "== $L"
  python tools/layer_table.py 2 96 2>/dev/null | head -5 | tail -4
  for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unet ms/step', d['ms_per_step'], 'k3pp TF', d['roofline']['achieved'])"; done
done
