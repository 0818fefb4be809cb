#!/bin/bash
# round-4 GPU session 24: the self-paired butterflies of the 8-point XL plan through the regular pair operations: A/B of the live
# session against the build before, then the whole GPU suite (the per-step kernels of that plan changed with it)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s24
VARIANTS="live_tw1 live_self live_tw1 live_self" bash tools/sessions/r4_session20.sh
cp gpurun_out/r4s20/variants.txt gpurun_out/r4s24/variants.txt
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider -x > gpurun_out/r4s24/pytest_gpu.log 2>&1
echo "gpu rc=$?"; tail -5 gpurun_out/r4s24/pytest_gpu.log
