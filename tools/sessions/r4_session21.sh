#!/bin/bash
# round-4 GPU session 21: live sessions with rows 0 of the pass twiddles in LDS (the other powers formed in registers): the live
# tests, then the per-step time against the build before (global table loads)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s21
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu --timeout 300 -p no:cacheprovider -k "live" > gpurun_out/r4s21/pytest_live.log 2>&1
echo "live rc=$?" ; tail -3 gpurun_out/r4s21/pytest_live.log
VARIANTS="live_glob live_tw1 live_glob live_tw1" bash tools/sessions/r4_session20.sh
cp gpurun_out/r4s20/variants.txt gpurun_out/r4s21/variants.txt
