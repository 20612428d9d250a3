#!/bin/bash
# accumulate-mode read-back prefetched under the MFMAs (k3pp_kernel<3>): tests on the new build, then old vs new library, same box
mkdir -p gpurun_out
L=medicalsemseg_amd
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline.py tests/test_gpu_engine.py -x -q -k "accumulate or sliding or split" > gpurun_out/r3c_k3pp3_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3c_k3pp3_pytest.log
[ $rc -ne 0 ] && exit $rc
for r in 1 2 3; do
  for v in old new; do
    MSSEG_LIB=$PWD/$L/libmsseg_hip_$v.so python bench.py --workload sliding_window --no-cpu-baseline --steps 5 --all-groups 2>/dev/null > gpurun_out/r3c_k3pp3_${v}_$r.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r3c_k3pp3_${v}_$r.json"))
g=[x for x in d["roofline"]["groups"] if x["group"]=="conv3d_k3_fwd/v3"][0]
print("$v r$r", d["value"], "vol/s; conv3d_k3_fwd/v3 kernel avg us", round(g["kernel_avg_ms"]*1e3,1), "x", g["launches_per_step"])
PY
  done
done
