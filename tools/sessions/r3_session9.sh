#!/bin/bash
# round-3 GPU session 9: rocprofv3 kernel-trace stats + PMC passes for the three kernels the bench line quotes
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
PROF_PASSES=8 bash tools/profile_gpu.sh r3_batch > gpurun_out/r3s9_batch.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r3_stream --mode stream > gpurun_out/r3s9_stream.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r3_chain --filter chain --chunk 8192 --fs 96000 > gpurun_out/r3s9_chain.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r3_config4 --filter highcut --channels 8192 > gpurun_out/r3s9_config4.log 2>&1
for t in r3_batch r3_stream r3_chain r3_config4; do echo "=== $t"; head -12 gpurun_out/prof_$t/summary.txt; grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAVES|SQ_INSTS_VALU|SQ_WAIT_ANY|SQ_WAVE_CYCLES|GRBM_GUI" gpurun_out/prof_$t/summary.txt; done
