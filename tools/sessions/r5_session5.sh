#!/bin/bash
# round-5 GPU session 5: (1) config 2's per-chunk transform without the cross-lane pairing (plan variants 22 / 24 / 25: M = 4096 with 32
# points per thread in two waves, 128 VGPRs: eight transforms per CU) - checked, then A/B in stream mode; (2) the chain on variant 13 again
# (512 threads x 32 points) on this round's tree; (3) the live figures again now that the bench closes its engines.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s5
mkdir -p $O
for v in 22 24 25; do ADSP_PLAN_VARIANT=$v timeout 300 python tools/check_variant.py 4096 2 2>&1 | grep -v "^$" | tee -a $O/check_variants.txt; done
S="python bench.py --mode stream --no-cpu-baseline --no-latency --no-graph --steps 2048 --warmup 512 --runs 3"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], r["ms_per_step"], r["shader_mhz"])'
for v in "" 22 24 25 ""; do
  for p in 1 2; do
    echo "stream pipeline=$p variant=[$v] $(ADSP_PLAN_VARIANT=$v timeout 300 $S --pipeline $p 2>/dev/null | python -c "$pick")" | tee -a $O/stream_variants.txt
  done
done
# the same plans in batch mode as 2N transforms (is the two-wave form faster per transform?)
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3 --fft-mult 2"
pickb='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["roofline"]["avg_launch_us"], r["shader_mhz"], d.get("max_rel_err"))'
for v in "" 22 24; do echo "batch 2N variant=[$v] $(ADSP_PLAN_VARIANT=$v timeout 300 $B 2>/dev/null | python -c "$pickb")" | tee -a $O/stream_variants.txt; done
C="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3 --filter chain --chunk 8192 --fs 96000"
for v in "" 13 ""; do echo "chain variant=[$v] $(ADSP_PLAN_VARIANT=$v timeout 300 $C 2>/dev/null | python -c "$pickb")" | tee -a $O/chain_variants.txt; done
/usr/bin/time -v timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench(default) rc=$?"; grep -E "Elapsed|Maximum resident" $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5s5/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "parity", d["max_rel_err"], "oracle", d["oracle_check"]["max_rel_err"])
    for k, c in d.get("configs", {}).items(): print("  ", k, c.get("value"), c["roofline"]["frac"], c["roofline"]["traffic"])
    c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
    lv = c3.get("resident_live", {})
    print("  config3", c3.get("us_per_step"), c3.get("launch_per_step", {}).get("us_per_step"), "live", lv.get("stream_producer", {}).get("us_per_step"), lv.get("host_producer", {}).get("us_per_step"))
    print("  numpy_api", json.dumps(d["latency"]["numpy_api"])[:500])
except Exception as e:
    print("no line:", e)
PY
