#!/bin/bash
# round-5 GPU session 13 (NEGATIVE RESULT - the two-block kernel is not in the tree; profiles/r5_upols_two_blocks_per_workgroup.txt): two consecutive output blocks of a channel per workgroup of the multiply launch (a partition's table entries
# loaded once for both) against one block per workgroup (ADSP_UPOLS_BLOCKS_PER_WG=1), alternating on one box, with the round's first form
# of the kernel (build_ab/libadsp_mac_old.so) as the anchor; the long-kernel tests in both forms.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s13
mkdir -p $O
for r in 1 2; do
  for per in 2 1; do
    echo "== blocks per workgroup: $per" | tee -a $O/upols_ab.txt
    ADSP_UPOLS_BLOCKS_PER_WG=$per timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt
  done
done
echo "== anchor: the kernel of session 11's 'old'" | tee -a $O/upols_ab.txt
ADSP_LIB=$PWD/build_ab/libadsp_mac_old.so timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt
for per in 2 1; do
  ADSP_UPOLS_BLOCKS_PER_WG=$per timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition" 2>&1 | tail -5 | tee -a $O/tests.txt
done
