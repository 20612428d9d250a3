#!/bin/bash
# same-box A/B of a bench workload: tools/r3_ab_swin.sh <tag> <workload> "<ENV=1 ...>" ["<ENV ...>" ...]   ("-" = no extra environment)
tag=$1; wl=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out
for round in 1 2; do
  i=0
  for envs in "$@"; do
    i=$((i+1))
    [ "$envs" = "-" ] && envs=""
    env $envs python bench.py --workload $wl --no-cpu-baseline --all-groups --steps 10 --warmup 3 > $out/v${i}_r$round.json 2> $out/v${i}_r$round.err || tail -3 $out/v${i}_r$round.err
  done
done
python3 - "$out" "$@" <<'PY'
import json, sys
out, variants = sys.argv[1], sys.argv[2:]
for i, v in enumerate(variants, 1):
    ms = []
    for r in (1, 2):
        try:
            d = json.load(open(f"{out}/v{i}_r{r}.json")); ms.append(d["ms_per_step"])
        except Exception as e:
            ms.append(None)
    print(f"variant {i} [{v}]: ms_per_step {ms}")
d = json.load(open(f"{out}/v1_r2.json"))
for g in d["roofline"]["groups"][:28]:
    print(f"  {g['group']:36s} {g['launches_per_step']:5.1f}/step  share {g['share_of_step']:.3f}  entry {g['avg_ms']*1e3:7.1f} us  kernel {('%.1f' % (g['kernel_avg_ms']*1e3)) if g['kernel_avg_ms'] else '   -'} us  {g['achieved']} {g['unit']}")
PY
