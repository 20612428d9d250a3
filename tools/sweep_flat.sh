#!/bin/bash
for n in 2 3; do
  export MSSEG_WG_FLAT_PER_CU=$n
  for i in 1 2; do python bench.py --workload swin_unetr --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat_per_cu=$n', d['ms_per_step'])"; done
done
