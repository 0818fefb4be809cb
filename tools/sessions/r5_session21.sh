#!/bin/bash
# round-5 GPU session 21: the bench line's long-kernel block with the numpy-API latency of Example4's own call (one mono chunk of 88200 samples); its test.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5s21
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "long_kernel_figures or smoke or raw_abi" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r5s21/bench_default.json 2> gpurun_out/r5s21/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r5s21/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic']); print(json.dumps(d['latency']['long_kernels'])[:900])"
