#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <tag>   (expects medicalsemseg_amd/libmsseg_hip_old.so / _new.so)
tag=$1; out=gpurun_out/$tag; mkdir -p $out
L=medicalsemseg_amd
for round in 1 2; do
  for v in old new; do
    cp $L/libmsseg_hip_$v.so $L/libmsseg_hip.so
    for w in fwd fwdstats; do timeout -k 10 120 python tools/bench_conv.py $w 32 32 96 20 2>&1 | tail -1 | sed "s/^/$v r$round B2 /"; done
    MSSEG_BENCH_N=8 timeout -k 10 120 python tools/bench_conv.py fwdstats 32 32 96 20 2>&1 | tail -1 | sed "s/^/$v r$round B8 /"
    python bench.py --no-cpu-baseline --no-sliding-window --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v r$round unet step ms', d['ms_per_step'])"
    python bench.py --workload sliding_window --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v r$round sliding window vol/s', d['value'])"
  done
done
cp $L/libmsseg_hip_new.so $L/libmsseg_hip.so
