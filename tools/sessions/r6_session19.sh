#!/bin/bash
# round-6 GPU session 19: the reference's own harness (ModuleTests.py's ten loops) through compat.install() against kat_moduletests.npz
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s19
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_moduletests.py -q -m gpu -rf 2>&1 | tail -40 | tee $O/pytest_moduletests.log
