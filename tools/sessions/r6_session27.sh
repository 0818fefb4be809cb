#!/bin/bash
# round-6 GPU session 27: soak of the final kernels - the randomised differential test at ten times its seeded cases (ADSP_FUZZ_SCALE=10: 2000 geometries / call patterns
# against the float64 direct sum, every sample of every channel)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s27
mkdir -p $O
ADSP_FUZZ_SCALE=10 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error|skipped" | tail -8 | tee $O/fuzz_x10.txt
