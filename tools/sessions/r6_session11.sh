#!/bin/bash
# round-6 GPU session 11: LDS exchange addresses split by hand (abl/prev.so = the tree before, abl/old.so = rounds 1 - 5), parity first.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s11
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round5.py -x -q -m gpu > $O/pytest_parity.log 2>&1
echo "pytest(parity+fuzz+round5) rc=$?"; tail -2 $O/pytest_parity.log
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 4"
ab() {  # ab "<bench args>" lib...
  args=$1; shift
  for r in 1 2 3; do for l in "$@"; do
    if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
    echo "$l $(ADSP_LIB=$lib timeout 300 $B $args 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))')"
  done; done
}
echo "== headline (config 2 batch)" | tee $O/ab.txt
ab "" old prev default 2>&1 | tee -a $O/ab.txt
echo "== chain (config 5)" | tee -a $O/ab.txt
ab "--filter chain --chunk 8192 --fs 96000" old prev default 2>&1 | tee -a $O/ab.txt
echo "== N = 2048 batch (M = 4096 two-wave plan)" | tee -a $O/ab.txt
ab "--chunk 2048 --channels 8192" old prev default 2>&1 | tee -a $O/ab.txt
echo "== N = 1024 batch (M = 2048 one-wave plan)" | tee -a $O/ab.txt
ab "--chunk 1024 --channels 16384" old prev default 2>&1 | tee -a $O/ab.txt
echo "== EQ, N = 4096 batch (complex spectrum)" | tee -a $O/ab.txt
ab "--filter eq3" old prev default 2>&1 | tee -a $O/ab.txt
echo "== config 3 shape per step (EQ, N = 512, 4096 ch), stream mode" | tee -a $O/ab.txt
ab "--filter eq3 --chunk 512 --mode stream" old prev default 2>&1 | tee -a $O/ab.txt
echo "== config 2 per chunk (stream)" | tee -a $O/ab.txt
ab "--mode stream" old prev default 2>&1 | tee -a $O/ab.txt
