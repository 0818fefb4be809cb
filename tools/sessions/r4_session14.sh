#!/bin/bash
# round-4 GPU session 14: rocprofv3 kernel-trace stats + PMC passes for the kernels the bench line quotes (final kernel sources),
# and the kernel trace of live sessions
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
PROF_PASSES=8 bash tools/profile_gpu.sh r4_batch > gpurun_out/r4s14_batch.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r4_stream --mode stream --pipeline 1 > gpurun_out/r4s14_stream.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r4_chain --filter chain --chunk 8192 --fs 96000 > gpurun_out/r4s14_chain.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r4_config4 --filter highcut --channels 8192 > gpurun_out/r4s14_config4.log 2>&1
for t in r4_batch r4_stream r4_chain r4_config4; do echo "=== $t"; head -8 gpurun_out/prof_$t/summary.txt | cut -c1-200; grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAVES|SQ_INSTS_VALU |SQ_WAIT_ANY|SQ_WAVE_CYCLES|GRBM_GUI" gpurun_out/prof_$t/summary.txt; done
mkdir -p gpurun_out/prof_r4_live
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4_live/trace -o t -- python $GRAFT_REPO_ROOT/tools/live_trace.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r4_live/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof_r4_live/run.log
f=$(find gpurun_out/prof_r4_live/trace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-220
