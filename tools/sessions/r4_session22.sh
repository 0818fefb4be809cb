#!/bin/bash
# round-4 GPU session 22: live sessions - the chunk fetched ahead is unpacked before the step's stores, the lane index laundered per
# step (no spills): live tests, then A/B against the build before
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s22
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu --timeout 300 -p no:cacheprovider -k "live" > gpurun_out/r4s22/pytest_live.log 2>&1
echo "live rc=$?" ; tail -3 gpurun_out/r4s22/pytest_live.log
export ADSP_LIVE_TRACE_RAW=1
VARIANTS="${AB:-live_tw1 live_unpack live_tw1 live_unpack}" bash tools/sessions/r4_session20.sh
cp gpurun_out/r4s20/variants.txt gpurun_out/r4s22/variants.txt
grep -B8 "live trace" gpurun_out/r4s22/variants.txt | tail -20
