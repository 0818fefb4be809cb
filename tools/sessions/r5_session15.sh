#!/bin/bash
# round-5 GPU session 15: the accumulation folded into the multiply-adds of the pair operation (256 instead of 320 vector instructions per
# partition and wave in the multiply launch) against the tree before it (build_ab/libadsp_head.so), alternating on one box, both block sizes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s15
mkdir -p $O
for r in 1 2; do
  for lib in "" build_ab/libadsp_head.so; do
    for b in 8192 16384; do
      echo "== lib=[${lib:-product}] block $b" | tee -a $O/upols_ab.txt
      if [ -z "$lib" ]; then timeout 300 python tools/bench_upols.py --only upols --block $b 2>&1 | tail -1 | tee -a $O/upols_ab.txt
      else ADSP_LIB=$PWD/$lib timeout 300 python tools/bench_upols.py --only upols --block $b 2>&1 | tail -1 | tee -a $O/upols_ab.txt; fi
    done
  done
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition" 2>&1 | tail -5 | tee $O/tests.txt
