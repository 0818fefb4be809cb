#!/bin/bash
# round-6 GPU session 25: host calls of the long-kernel engine (a mono chunk of 88200 samples through numpy: Example4's shape) through the pinned window
# against the staging copies (ADSP_UPOLS_HOST_STAGED=1), alternating; then the long-kernel tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s25
mkdir -p $O
for r in 1 2; do
  echo "== window" | tee -a $O/ab.txt; timeout 300 python tools/probe_long_kernel_1ch.py 2>/dev/null | grep -E "host us|apply us|apply_host us" | tee -a $O/ab.txt
  echo "== staged" | tee -a $O/ab.txt; ADSP_UPOLS_HOST_STAGED=1 timeout 300 python tools/probe_long_kernel_1ch.py 2>/dev/null | grep -E "host us|apply us|apply_host us" | tee -a $O/ab.txt
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_parity.py -q -m gpu -k "long or upols or 88200 or partition" -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -5 | tee $O/pytest_subset.txt
