#!/bin/bash
# per-layer sweep of the generic conv k3 kernel's tile / cout-block choice on the small grids
for f in none 1,32 1,16 2,32 2,16; do
  if [ $f = none ]; then unset MSSEG_K3_FORCE; else export MSSEG_K3_FORCE=$f; fi
  echo "== MSSEG_K3_FORCE=$f"
  python tools/layer_table.py 2 96 2>/dev/null | tail -9 | awk '{print $1, $2, "fwd", $6, "dgrad", $9}'
done
