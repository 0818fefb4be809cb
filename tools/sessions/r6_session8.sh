#!/bin/bash
# round-6 GPU session 8: butterflies with fused constant twiddles (6 multiply-adds per radix-2 combine, pass twiddles folded into the DFT's
# first level) and the lane-pair exchange of the 16-byte accesses as v_cndmask_b32_dpp - parity of the product build, then A/B against
# the tuning builds without one / both (abl/old.so = rounds 1 - 5 instruction selection), alternating on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_parity.log 2>&1
echo "pytest(parity+fuzz) rc=$?"; tail -3 $O/pytest_parity.log
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 4"
ab() {  # ab "<bench args>" lib...
  args=$1; shift
  for r in 1 2 3; do for l in "$@"; do
    if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
    echo "$l $(ADSP_LIB=$lib timeout 300 $B $args 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))')"
  done; done
}
echo "== headline (config 2 batch)" | tee $O/ab.txt
ab "" old fused dpp default 2>&1 | tee -a $O/ab.txt
echo "== chain (config 5)" | tee -a $O/ab.txt
ab "--filter chain --chunk 8192 --fs 96000" old fused dpp default 2>&1 | tee -a $O/ab.txt
echo "== N = 2048 batch (M = 4096 two-wave plan)" | tee -a $O/ab.txt
ab "--chunk 2048 --channels 8192" old default 2>&1 | tee -a $O/ab.txt
echo "== config 2 per chunk (stream, XL plan)" | tee -a $O/ab.txt
ab "--mode stream" old default 2>&1 | tee -a $O/ab.txt
