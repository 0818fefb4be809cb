#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s17; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -x 2>&1 | tail -30 ) 2>&1 | tee $O/contract.log
