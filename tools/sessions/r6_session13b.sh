#!/bin/bash
# round-6 GPU session 13b: M = 16384 on 32 points per thread in 512 threads as the DEFAULT plan (config 5, EQ at N = 8192), with the window's
# head and tail on plain loads (the hybrid of the 64-point plan, now keyed on M >= 16384) - whole suite, then A/B against rounds 1 - 5's
# selection (abl/old.so: 64-point plan) and against every window load non-temporal (ADSP_NT_HYBRID=0).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s13b
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf -x > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 4"
run() { echo "$1 $(env $2 timeout 300 $B $3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))')"; }
echo "== chain (config 5)" | tee $O/ab.txt
for r in 1 2 3; do
  run old ADSP_LIB=abl/old.so "--filter chain --chunk 8192 --fs 96000" | tee -a $O/ab.txt
  run default X=1 "--filter chain --chunk 8192 --fs 96000" | tee -a $O/ab.txt
  run all_nt ADSP_NT_HYBRID=0 "--filter chain --chunk 8192 --fs 96000" | tee -a $O/ab.txt
done
echo "== EQ at N = 8192, batch" | tee -a $O/ab.txt
for r in 1 2; do
  run old ADSP_LIB=abl/old.so "--filter eq3 --chunk 8192 --fs 96000" | tee -a $O/ab.txt
  run default X=1 "--filter eq3 --chunk 8192 --fs 96000" | tee -a $O/ab.txt
  run all_nt ADSP_NT_HYBRID=0 "--filter eq3 --chunk 8192 --fs 96000" | tee -a $O/ab.txt
done
echo "== long kernels (unchanged plans; check)" | tee -a $O/ab.txt
timeout 300 python tools/bench_upols.py --only upols 2>/dev/null | tail -1 | cut -c1-900 | tee -a $O/ab.txt
