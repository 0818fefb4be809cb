#!/bin/bash
# round-5 GPU session 11: the multiply-accumulate kernel of the uniformly partitioned engines as one stream of stages requested four ahead
# (buffer loads, no branch inside a partition, thread 0's self-paired butterflies by 17 lanes in front of the loop) against the previous
# form (build_ab/libadsp_mac_old.so = HEAD~ of adsp_upols.hip), alternating on one box; then the long-kernel tests and the two new tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s11
mkdir -p $O
for r in 1 2; do
  for lib in "" build_ab/libadsp_mac_old.so; do
    echo "== lib=[${lib:-product}]" | tee -a $O/upols_ab.txt
    if [ -z "$lib" ]; then timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt
    else ADSP_LIB=$PWD/$lib timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt; fi
  done
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or rides_a_session or long_kernel or partition" 2>&1 | tail -5 | tee $O/tests.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/upols_kernel_stats.csv
rm -rf $O/prof
head -8 $O/upols_kernel_stats.csv | cut -c1-200
