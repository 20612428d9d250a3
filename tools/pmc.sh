#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...> -- <bench_conv args>   (counter pass only: no stats/trace domains)
tag=$1; shift
ctr=()
while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag/run.log 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/$tag/*/*_counter_collection.csv')
if not f:
    print(open('gpurun_out/$tag/run.log').read()[-2000:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'igemm' in k or 'instnorm' in k or 'stats' in k or 'k3pp' in k or 'k3wg' in k or 'k3c48' in k:
        print(k, {c: sum(v)/len(v) for c, v in d.items()}, 'launches', len(list(d.values())[0]))
print(open('gpurun_out/$tag/run.log').read()[-300:])
PY
