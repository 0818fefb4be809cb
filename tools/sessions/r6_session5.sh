#!/bin/bash
# round-6 GPU session 5: the whole suite on the split library; what bounds config 3's launch-per-step kernel: the
# per-step launches with pass twiddles read from two addresses (abl1), pair tables not loaded (abl2), both (abl3), butterflies replaced
# by copies (abl32) against the same tuning build with nothing removed (abl0) - kernel us per launch from HIP events.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s5
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -rf 2>&1 | tail -12 | tee $O/tests_all.txt
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"], d["roofline"]["frac"])'
S="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --no-parity-check --mode stream --pipeline 1 --no-graph --filter eq3 --chunk 512 --channels 4096 --steps 4096 --warmup 512 --runs 3"
for r in 1 2; do for l in abl0 abl1 abl2 abl3 abl32; do
  echo "config3 per-step $l $(ADSP_BENCH_NO_SANITY=1 ADSP_LIB=$PWD/abl/$l.so timeout 300 $S 2>/dev/null | python -c "$pick")" | tee -a $O/config3_per_step_bounds.txt
done; done
echo "config3 per-step product $(timeout 300 $S 2>/dev/null | python -c "$pick")" | tee -a $O/config3_per_step_bounds.txt
