import csv, glob, sys, re
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
log = open(d + '/run.log').read()
m = re.search(r'"steps": (\d+), "warmup": (\d+)', log)
nst = (int(m.group(1)) + max(int(m.group(2)), 1)) if m else 1
out = []
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    n = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)[:70]
    out.append(f"{float(r['TotalDurationNs'])/1e6/nst:8.3f} ms/step {int(r['Calls'])/nst:6.1f} calls avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%  {n}")
out.append(f"GPU busy per step: {tot/1e6/nst:.3f} ms over {nst} steps")
m = re.search(r'\{"metric".*\}', log)
if m: out.append(m.group(0)[:1500])
open(d + '/summary.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
