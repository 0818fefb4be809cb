#!/bin/bash
# round-5 GPU session 16 (NEGATIVE RESULT - the side-stream form is not in the tree; profiles/r5_upols_ring_copy_side_stream.txt): the long-kernel engines' ring update on a side stream beside the multiply launch against the copies behind the
# kernels on the caller's stream (ADSP_UPOLS_RING_COPY=inline), alternating on one box; the long-kernel tests and the bench figure's test.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s16
mkdir -p $O
for r in 1 2; do
  for mode in side inline; do
    echo "== ring update: $mode" | tee -a $O/upols_ab.txt
    ADSP_UPOLS_RING_COPY=$mode timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition" 2>&1 | tail -5 | tee $O/tests.txt
