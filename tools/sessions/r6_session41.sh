#!/bin/bash
# round-6 GPU session 41: soak after session 39's one abort (not reproduced in session 40): the randomised long-kernel test 40 times over, the long-kernel tests of rounds 5 / 6
# ten times, the whole suite three times - on one box, every failure kept.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s41
mkdir -p $O
for i in $(seq 1 40); do timeout 300 python -m pytest tests/test_gpu_round5.py -q -m gpu -p no:cacheprovider -k "randomised" 2>&1 | tail -1; done | sort | uniq -c | tee $O/randomised_x40.txt
for i in $(seq 1 10); do timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -q -m gpu -p no:cacheprovider -k "upols or long or Upols" 2>&1 | tail -1 | sed 's/ in [0-9.]*s.*//'; done | sort | uniq -c | tee $O/long_kernel_x10.txt
for i in 1 2 3; do timeout 1200 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > $O/all_$i.log 2>&1; echo "rc=$?"; tail -1 $O/all_$i.log; done | tee $O/all_x3.txt
grep -l "Fatal\|FAILED" $O/all_*.log
