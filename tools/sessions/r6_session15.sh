#!/bin/bash
# round-6 GPU session 15: the long-kernel launches at 64 points per thread (blocks of 16384) - the one place such a plan still runs - with
# the laundered lane index (now in the tree), and with the split exchange addresses / the DPP selects on top (tuning builds); 1024 and 64
# channels, alternating.  Then the chain and the headline once more on the final lane-index choice per plan.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s15
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_part.log 2>&1
echo "pytest(parity, round5, round6) rc=$?"; tail -1 $O/pytest_part.log
echo "== long kernels: us per call" | tee $O/ab.txt
for r in 1 2 3; do for l in old default p64hand p64dpp p64both; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "$l $(ADSP_LIB=$lib timeout 300 python tools/bench_upols.py --only upols 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: v["upols"]["us_per_call"] for k, v in d.items()})')" | tee -a $O/ab.txt
done; done
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 4"
ab() {  # ab "<bench args>" lib...
  args=$1; shift
  for r in 1 2 3; do for l in "$@"; do
    if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
    echo "$l $(ADSP_LIB=$lib timeout 300 $B $args 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))')"
  done; done
}
echo "== chain (config 5)" | tee -a $O/ab.txt
ab "--filter chain --chunk 8192 --fs 96000" old default 2>&1 | tee -a $O/ab.txt
echo "== headline" | tee -a $O/ab.txt
ab "" old default 2>&1 | tee -a $O/ab.txt
