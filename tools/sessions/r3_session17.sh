#!/bin/bash
# round-3 GPU session 17: ten times the randomised differential tests on the final tree
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s17; mkdir -p $O
export TMPDIR=/tmp
( ADSP_FUZZ_SCALE=10 timeout 1700 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -n 4 2>&1 | tail -8 ) > $O/fuzz.log 2>&1
echo "fuzz x10: $(grep -E 'passed|failed|error' $O/fuzz.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/fuzz.log | head
