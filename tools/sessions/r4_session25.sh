#!/bin/bash
# round-4 GPU session 25: the traffic passes again on the final kernel sources (the measured kernels' ISA is unchanged, the source stamp
# is not): kernel trace + FETCH_SIZE + WRITE_SIZE per config; WHICH="batch chain stream config4"
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
for w in ${WHICH:-batch chain}; do
  case $w in
    batch)   PROF_ONLY="4 5" bash tools/profile_gpu.sh r4g_batch > gpurun_out/r4s25_batch.log 2>&1;;
    chain)   PROF_ONLY="4 5" bash tools/profile_gpu.sh r4g_chain --filter chain --chunk 8192 --fs 96000 > gpurun_out/r4s25_chain.log 2>&1;;
    stream)  PROF_ONLY="4 5" bash tools/profile_gpu.sh r4g_stream --mode stream --pipeline 1 > gpurun_out/r4s25_stream.log 2>&1;;
    config4) PROF_ONLY="4 5" bash tools/profile_gpu.sh r4g_config4 --filter highcut --channels 8192 > gpurun_out/r4s25_config4.log 2>&1;;
  esac
  echo "=== $w"; head -4 gpurun_out/prof_r4g_$w/summary.txt | cut -c1-200; grep -E "FETCH_SIZE|WRITE_SIZE" gpurun_out/prof_r4g_$w/summary.txt
done
