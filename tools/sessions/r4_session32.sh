#!/bin/bash
# round-4 GPU session 32: the final tree once more - whole -m gpu suite and smoke()
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s32
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r4s32/pytest_gpu.log 2>&1
echo "suite rc=$?"; grep -n "passed\|failed" gpurun_out/r4s32/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4s32/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4s32/smoke.log
