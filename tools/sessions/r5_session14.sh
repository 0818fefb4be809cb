#!/bin/bash
# round-5 GPU session 14: the long-kernel engines on blocks of 16384 samples (the 64-points-per-thread plan: half the partitions, half the
# bytes the multiply launch reads per output sample) against blocks of 8192, alternating on one box; the long-kernel tests (both block
# sizes, randomised shapes); per-kernel times of both.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s14
mkdir -p $O
for r in 1 2; do
  for b in 16384 8192; do
    echo "== block $b" | tee -a $O/upols_ab.txt
    timeout 300 python tools/bench_upols.py --only upols --block $b 2>&1 | tail -1 | tee -a $O/upols_ab.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition" 2>&1 | tail -5 | tee $O/tests.txt
cd /tmp
for b in 16384 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$b -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --block $b --calls 8 --channels 1024 > /dev/null 2>&1
  find $GRAFT_REPO_ROOT/$O/prof$b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/upols_kernel_stats_b$b.csv
  rm -rf $GRAFT_REPO_ROOT/$O/prof$b
  head -4 $GRAFT_REPO_ROOT/$O/upols_kernel_stats_b$b.csv | cut -c1-220
done
