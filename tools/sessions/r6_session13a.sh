#!/bin/bash
# round-6 GPU session 13a: the whole suite after the M = 3 * 2^k fix of the hand-split write addresses; then the chain (config 5) on the
# 512-thread / 32-points-per-thread form of the M = 16384 transform (plans_var.hip entry 13) against the 64-point default - the 32-point
# plans gained two to three times as much from this round's instruction-level rework as the 64-point one.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s13a
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf -x > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 4"
echo "== chain (config 5): tuning build of the tree, default plan / variant 13" | tee $O/ab.txt
ADSP_LIB=abl/cur.so ADSP_PLAN_VARIANT=13 timeout 120 python tools/check_variant.py 8192 4 2>&1 | tail -2 | tee -a $O/ab.txt
for r in 1 2 3; do for v in "" 13; do
  echo "variant=$v $(ADSP_LIB=abl/cur.so ADSP_PLAN_VARIANT=$v timeout 300 $B --filter chain --chunk 8192 --fs 96000 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))')" | tee -a $O/ab.txt
done; done
echo "== EQ at N = 8192, batch (M = 16384, complex spectrum)" | tee -a $O/ab.txt
for r in 1 2; do for v in "" 13; do
  echo "variant=$v $(ADSP_LIB=abl/cur.so ADSP_PLAN_VARIANT=$v timeout 300 $B --filter eq3 --chunk 8192 --fs 96000 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))')" | tee -a $O/ab.txt
done; done
