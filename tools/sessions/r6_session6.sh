#!/bin/bash
# round-6 GPU session 6: blocks of 16384 on 32 points per thread in 512 threads (abl/upols_t512.so: 64 accumulators, four stages of
# requests ahead) against the 64-point plan of the tree (two stages ahead), alternating on one box; parity of the variant.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s6
mkdir -p $O
for r in 1 2; do for l in "" upols_t512; do
  echo "== lib=[${l:-product}] block 16384" | tee -a $O/upols_t512_ab.txt
  if [ -z "$l" ]; then timeout 300 python tools/bench_upols.py --only upols --block 16384 2>&1 | tail -1 | tee -a $O/upols_t512_ab.txt
  else ADSP_LIB=$PWD/abl/$l.so timeout 300 python tools/bench_upols.py --only upols --block 16384 2>&1 | tail -1 | tee -a $O/upols_t512_ab.txt; fi
done; done
ADSP_LIB=$PWD/abl/upols_t512.so timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu -k "upols" -rf 2>&1 | tail -8 | tee $O/tests_t512.txt
