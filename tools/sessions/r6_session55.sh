#!/bin/bash
# round-6 GPU session 55: last checks on the final library - the randomised differential test at ten times its cases, the long-kernel tests five times over, the examples
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s55
mkdir -p $O
ADSP_FUZZ_SCALE=10 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed" | tail -1 | tee $O/fuzz_x10.txt
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed" | tail -1; done | tee $O/long_x5.txt
for e in examples/*.py; do echo "== $e"; timeout 300 python $e > $O/$(basename $e .py).log 2>&1; echo "rc=$?"; tail -2 $O/$(basename $e .py).log | cut -c1-200; done
