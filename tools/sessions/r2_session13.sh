#!/bin/bash
# round-2 GPU session 13: randomised differential test (FFT engines vs the exact engine), then the whole suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s13; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -40 ) > $O/fuzz.log 2>&1
grep -E "passed|failed|error|^FAILED|^E  " $O/fuzz.log | head -40
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
echo "suite: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"
