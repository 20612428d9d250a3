#!/bin/bash
# A/B of k3pp builds on one conv shape, with the in-kernel role counters
for L in "$@"; do
  export MSSEG_LIB=$PWD/$L
  echo "== $L"
  for m in fwd fwdstats; do
    python tools/bench_conv.py $m 32 32 96 50 2>/dev/null | tail -1
    MSSEG_K3PP_TIMING=1 python tools/bench_conv.py $m 32 32 96 50 2>/dev/null | tail -2
  done
done
