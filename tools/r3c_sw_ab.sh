#!/bin/bash
# sliding window: K-split of the 24^3 level on / off, three rounds, 5 volumes each, all groups
mkdir -p gpurun_out/r3c_sw_ab
for r in 1 2 3; do
  MSSEG_NO_KSPLIT_INFER=1 python bench.py --workload sliding_window --no-cpu-baseline --steps 5 --all-groups > gpurun_out/r3c_sw_ab/off_$r.json 2> /dev/null
  python bench.py --workload sliding_window --no-cpu-baseline --steps 5 --all-groups > gpurun_out/r3c_sw_ab/on_$r.json 2> /dev/null
done
python3 - <<'PY'
import json
for k in ("off", "on"):
    vals = [json.load(open(f"gpurun_out/r3c_sw_ab/{k}_{r}.json"))["value"] for r in (1, 2, 3)]
    print(k, vals)
for k in ("off", "on"):
    d = json.load(open(f"gpurun_out/r3c_sw_ab/{k}_3.json"))
    print(k)
    for g in d["roofline"]["groups"][:14]:
        print(f"  {g['group']:34s} {g['launches_per_step']:7.1f}/vol entry {g['avg_ms']*1e3:7.1f} us tot {g['avg_ms']*g['launches_per_step']:7.1f} ms kernel {g['kernel_avg_ms'] and round(g['kernel_avg_ms']*1e3,1)}")
PY
