#!/bin/bash
# round-6 GPU session 3: round-6 tests + the whole suite on the tree with the tuning split, the condition-variable host pipeline and
# the long-kernel engines' filter broadcast; BOUNDS by ablation (tuning flavour, wrong results): the chain's half-buffer exchanges
# forward 0 -> 1 and last inverse without their workgroup barriers (abl8192 against abl0), the long-kernel multiply launch without
# table traffic / without spectrum traffic / without both (upols_abl1 / 2 / 3); config 2 per chunk with rings of 3 / 12 / 48 slots
# (does the half chunk of history that FETCH_SIZE reports come from HBM?) and which memory-side counters this rocprofv3 offers.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests_round6.txt
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["runs"]["shader_mhz"])'
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --no-parity-check --steps 8 --warmup 3 --runs 3 --filter chain --chunk 8192 --fs 96000"
for r in 1 2; do for l in abl0 abl8192; do
  echo "chain $l $(ADSP_BENCH_NO_SANITY=1 ADSP_LIB=$PWD/abl/$l.so timeout 300 $B 2>/dev/null | python -c "$pick")" | tee -a $O/chain_wave_private_bound.txt
done; done
for r in 1 2; do for l in "" upols_abl1 upols_abl2 upols_abl3; do
  echo "== lib=[${l:-product}]" | tee -a $O/upols_ablations.txt
  if [ -z "$l" ]; then timeout 300 python tools/bench_upols.py --only upols --block 8192 2>&1 | tail -1 | tee -a $O/upols_ablations.txt
  else ADSP_LIB=$PWD/abl/$l.so timeout 300 python tools/bench_upols.py --only upols --block 8192 2>&1 | tail -1 | tee -a $O/upols_ablations.txt; fi
done; done
rocprofv3 --list-avail 2>/dev/null | grep -i -E "dram|EA0_RDREQ|EA_RDREQ|MALL|HBM|TCC_EA" | head -60 > $O/avail_memory_counters.txt; wc -l $O/avail_memory_counters.txt
S="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --mode stream --pipeline 1 --steps 768 --warmup 384 --runs 3"
for slots in 3 12 48; do
  echo "stream ring_slots=$slots $(timeout 300 $S --ring-slots $slots 2>/dev/null | python -c "$pick")" | tee -a $O/stream_ring_slots.txt
done
for slots in 3 48; do
  PROF_ONLY="4 5 6" PROF_PASSES=6 timeout 600 bash tools/profile_gpu.sh r6_stream_slots$slots --mode stream --pipeline 1 --ring-slots $slots > $O/prof_stream_slots$slots.log 2>&1
  cp gpurun_out/prof_r6_stream_slots$slots/summary.txt $O/stream_slots${slots}_summary.txt 2>/dev/null
done
grep -E "FETCH|WRITE|TCC" $O/stream_slots*_summary.txt | cut -c1-160
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/tests_all.txt
