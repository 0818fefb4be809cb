#!/bin/bash
# round-6 GPU session 26: M = 16384 on 512 threads with the STORE side's thread index formed anew (ADSP_FRESH_STORE=1: the kernel's only spill - 8 bytes per lane - gone)
# against the laundered copy kept to the end (=0), tuning builds of plans_f32.hip, alternating on one box: config 5 (chain), the EQ at N = 8192, the headline as a control
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s26
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 4"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"), d.get("max_rel_err"))'
echo "# library, Msamples/s, kernel us per launch, fraction, shader MHz, error of the timed output" | tee $O/ab.txt
for r in 1 2 3; do for l in fs0 fs1; do
  echo "chain $l $(ADSP_LIB=abl/$l.so timeout 300 $B --filter chain --chunk 8192 --fs 96000 2>/dev/null | python -c "$P")" | tee -a $O/ab.txt
done; done
for r in 1 2; do for l in fs0 fs1; do
  echo "eq8192 $l $(ADSP_LIB=abl/$l.so timeout 300 $B --filter eq3 --chunk 8192 --fs 96000 2>/dev/null | python -c "$P")" | tee -a $O/ab.txt
done; done
for l in fs0 fs1; do
  echo "headline $l $(ADSP_LIB=abl/$l.so timeout 300 $B 2>/dev/null | python -c "$P")" | tee -a $O/ab.txt
done
