#!/bin/bash
# round-6 GPU session 12: the whole suite on the tree with hand-split LDS addresses (32-point plans only); then the two-level twiddle
# threshold again under the new instruction balance: passes with S < 64 (tw64) / S < 32 (tw32) read all their powers from the table
# (8 loads per radix-16 butterfly instead of 3, no powers formed in registers) - tuning builds against the product library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s12
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
ab "" old default tw64 tw32 2>&1 | tee -a $O/ab.txt
echo "== chain (config 5)" | tee -a $O/ab.txt
ab "--filter chain --chunk 8192 --fs 96000" old default tw64 tw32 2>&1 | tee -a $O/ab.txt
echo "== EQ, N = 4096 batch (complex spectrum)" | tee -a $O/ab.txt
ab "--filter eq3" default tw64 2>&1 | tee -a $O/ab.txt
