#!/bin/bash
for d in 0 8 16 32; do
 for t in 128 256; do
  export MSSEG_K3WG_MINDIM=$d MSSEG_WG_TOTAL=$t
  echo "== MINDIM=$d WG_TOTAL=$t"
  python tools/layer_table.py 2 96 2>/dev/null | tail -9 | awk '{print $1, $2, $NF, $(NF-1), $(NF-2)}'
 done
done
