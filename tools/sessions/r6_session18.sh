#!/bin/bash
# round-6 GPU session 18: bounds on the final tree's headline kernel (ablation builds of the tuning flavour, wrong results by construction):
# 0 = nothing removed, 16384 = twiddle powers not formed in registers (the bound of keeping whole twiddle tables on chip),
# 24 = no global traffic at all (on-chip time), 8 = no window loads, 16 = no output stores.  Shader clock beside every run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s18
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --no-parity-check --steps 8 --warmup 4"
echo "# mask, Msamples/s, kernel us per launch, fraction, shader MHz" | tee $O/ablations.txt
for r in 1 2; do for m in 0 16384 24 8 16; do
  echo "abl$m $(ADSP_BENCH_NO_SANITY=1 ADSP_LIB=abl/abl$m.so timeout 300 $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"))')" | tee -a $O/ablations.txt
done; done
echo "# chain (config 5)" | tee -a $O/ablations.txt
for r in 1 2; do for m in 0 16384 24; do
  echo "abl$m $(ADSP_BENCH_NO_SANITY=1 ADSP_LIB=abl/abl$m.so timeout 300 $B --filter chain --chunk 8192 --fs 96000 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("shader_mhz"))')" | tee -a $O/ablations.txt
done; done
