#!/bin/bash
# Counter evidence for the two ping-pong kernels: matrix-pipe busy cycles, wave cycles, the clock.
# usage: tools/r3_pmc.sh <tag> <fwdstats|fwd|wgrad> <batch>      (counter passes only: --kernel-trace + --pmc)
tag=$1; what=$2; nb=${3:-8}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export MSSEG_BENCH_N=$nb
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py $what 32 32 96 30 > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; }
done
cd $GRAFT_REPO_ROOT && python3 - <<PY
import csv, glob, collections
out = 'gpurun_out/$tag'
res = []
for i in (1, 2, 3):
    cf = glob.glob(f'{out}/p{i}/*/*_counter_collection.csv'); kf = glob.glob(f'{out}/p{i}/*/*_kernel_trace.csv')
    if not cf or not kf:
        res.append(f'pass {i}: no output; ' + open(f'{out}/p{i}.log').read()[-400:]); continue
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kf[0])):
        dur[r['Kernel_Name'][:48]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cf[0])):
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'k3pp' in k or 'k3wg' in k:
            us = sorted(dur[k]); med = us[len(us) // 2]
            line = f'pass {i}  {k}  launches {len(us)}  median {med:.1f} us  ' + '  '.join(f'{c} {sum(v)/len(v):.4g}' for c, v in sorted(d.items()))
            g = d.get('GRBM_GUI_ACTIVE')
            if g:
                line += f'  | clock = GRBM_GUI_ACTIVE/8/wall = {sum(g)/len(g)/8/med/1e3:.3f} GHz'
            res.append(line)
    res.append(open(f'{out}/p{i}.log').read().strip().splitlines()[-1])
open(f'{out}/summary.txt', 'w').write('\n'.join(res) + '\n')
print('\n'.join(res))
PY
