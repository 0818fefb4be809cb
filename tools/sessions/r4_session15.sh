#!/bin/bash
# round-4 GPU session 15: kernel trace of live sessions (config 3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r4_live
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4_live/trace -o t -- python $GRAFT_REPO_ROOT/tools/live_trace.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r4_live/run.log 2>&1
cd $GRAFT_REPO_ROOT
grep "producer" gpurun_out/prof_r4_live/run.log
f=$(find gpurun_out/prof_r4_live/trace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-260
