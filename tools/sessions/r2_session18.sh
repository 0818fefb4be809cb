#!/bin/bash
# round-2 GPU session 18: rocprofv3 summaries of the final tree (headline batch, stream, chain), for profiles/
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s18; mkdir -p $O
export TMPDIR=/tmp
bash tools/profile_gpu.sh r2f_batch > $O/prof_batch.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r2f_stream --mode stream > $O/prof_stream.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r2f_chain --filter chain --chunk 8192 --fs 96000 > $O/prof_chain.log 2>&1
for t in batch stream chain; do echo "== $t"; grep -E "fftconv|SQ_INSTS_VALU|SQ_INSTS_VMEM|SQ_WAVES |FETCH_SIZE|WRITE_SIZE" gpurun_out/prof_r2f_$t/summary.txt | cut -c1-220; done
