#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k batchnorm > gpurun_out/r2r_t.log 2>&1; tail -2 gpurun_out/r2r_t.log
timeout -k 10 300 python bench.py --workload swin_unetr_official --no-cpu-baseline > gpurun_out/r2r_swo_graph.json 2> gpurun_out/r2r_swo_graph.err; cut -c1-220 gpurun_out/r2r_swo_graph.json; tail -3 gpurun_out/r2r_swo_graph.err
MSSEG_SWIN_NO_GRAPH=1 timeout -k 10 300 python bench.py --workload swin_unetr_official --no-cpu-baseline > gpurun_out/r2r_swo_eager.json 2> gpurun_out/r2r_swo_eager.err; cut -c1-220 gpurun_out/r2r_swo_eager.json
