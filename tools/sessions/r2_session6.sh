#!/bin/bash
# round-2 GPU session 6: profiles of the new large-transform plans (chain, N = 8192), final default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s6; mkdir -p $O
export TMPDIR=/tmp
PROF_PASSES=5 bash tools/profile_gpu.sh r2b_chain --filter chain --chunk 8192 --fs 96000 > $O/prof_chain.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r2b_lc8192 --chunk 8192 --channels 2048 > $O/prof_lc8192.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r2b_eq4096 --filter eq3 > $O/prof_eq4096.log 2>&1
for t in chain lc8192 eq4096; do echo "== $t"; grep -E "fftconv|SQ_INSTS_VALU|SQ_INSTS_VMEM|SQ_INSTS_LDS|SQ_WAVES |FETCH_SIZE|WRITE_SIZE|SQ_WAIT_INST_LDS|SQ_BUSY" gpurun_out/prof_r2b_$t/summary.txt | cut -c1-200; done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']); print(d['stream']); print(d['cpu_baseline']['value'], d['cpu_baseline']['variants_msamples_s'])"
python bench.py --filter chain --chunk 8192 --fs 96000 --no-cpu-baseline --no-latency > $O/bench_chain.json 2>> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_chain.json').read().strip().splitlines()[-1]); print('chain', d['value'], d['roofline'])"
