#!/bin/bash
# round-6 GPU session 14: the lane index taken anew from the hardware at every exchange (no scratch access left in the 32-point plans),
# laundering only where it paid (32 points per thread, >= 2 waves per transform) - whole suite, then A/B against the same tree with the
# laundered copy (abl/nolane.so) and rounds 1 - 5's selection (abl/old.so).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s14
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf -x > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 4"
ab() {  # ab "<bench args>" lib...
  args=$1; shift
  for r in 1 2 3; do for l in "$@"; do
    if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
    echo "$l $(ADSP_LIB=$lib timeout 300 $B $args 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))')"
  done; done
}
echo "== headline (config 2 batch)" | tee $O/ab.txt
ab "" old nolane default 2>&1 | tee -a $O/ab.txt
echo "== chain (config 5)" | tee -a $O/ab.txt
ab "--filter chain --chunk 8192 --fs 96000" old nolane default 2>&1 | tee -a $O/ab.txt
echo "== N = 2048 batch (M = 4096 two-wave plan)" | tee -a $O/ab.txt
ab "--chunk 2048 --channels 8192" old nolane default 2>&1 | tee -a $O/ab.txt
echo "== N = 1024 batch (M = 2048 one-wave plan)" | tee -a $O/ab.txt
ab "--chunk 1024 --channels 16384" old nolane default 2>&1 | tee -a $O/ab.txt
echo "== EQ, N = 4096 batch (complex spectrum)" | tee -a $O/ab.txt
ab "--filter eq3" old nolane default 2>&1 | tee -a $O/ab.txt
echo "== long kernels, old / nolane / default" | tee -a $O/ab.txt
for r in 1 2; do for l in old nolane default; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "$l $(ADSP_LIB=$lib timeout 300 python tools/bench_upols.py --only upols 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: v["upols"]["us_per_call"] for k, v in d.items()})')" | tee -a $O/ab.txt
done; done
