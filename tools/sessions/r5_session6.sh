#!/bin/bash
# round-5 GPU session 6: the two-wave M = 4096 plan as the DEFAULT: the whole -m gpu suite on it, the default bench line (stream figure!),
# A/B against the XL plan (variants 26 / 27) where M = 4096 is used, then the PMC passes again (plan_table.hpp changed: the stamp).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s6
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; tail -6 $O/pytest_all.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo "bench(default) rc=$?"; cat $O/bench_time.txt | tail -3
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5s6/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "parity", d["max_rel_err"], "oracle", d["oracle_check"]["max_rel_err"])
    for k, c in d.get("configs", {}).items(): print("  ", k, c.get("value"), c["roofline"]["frac"], c["roofline"]["traffic"])
    s = d["stream"]
    print("  stream", s.get("value"), s.get("us_per_step"), s.get("roofline_frac"), s.get("runs_us_per_step"), "one", s["one_stream"]["us_per_step"], s["one_stream"]["avg_kernel_us"], s["one_stream"].get("graph", {}).get("us_per_step"), "resident", s.get("resident", {}).get("us_per_step"))
    c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
    lv = c3.get("resident_live", {})
    print("  config3", c3.get("us_per_step"), c3.get("launch_per_step", {}).get("us_per_step"), "live", lv.get("stream_producer", {}).get("us_per_step"), lv.get("host_producer", {}).get("us_per_step"))
    print("  numpy_api", json.dumps(d["latency"]["numpy_api"])[:600])
except Exception as e:
    print("no line:", e)
PY
pickb='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], r["shader_mhz"], d.get("max_rel_err"))'
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3"
for shape in "--chunk 2048 --channels 8192" "--filter eq3 --chunk 2048 --channels 8192"; do
  for v in "" 27 ""; do echo "[$shape] variant=[$v] $(ADSP_PLAN_VARIANT=$v timeout 300 $B $shape 2>/dev/null | python -c "$pickb")" | tee -a $O/plan4096_ab.txt; done
done
for io in s16; do echo "[--io $io stream] $(timeout 300 python bench.py --io $io --mode stream --no-cpu-baseline --no-latency --no-graph --steps 1024 --warmup 256 --runs 3 --pipeline 1 2>/dev/null | python -c "$pickb")" | tee -a $O/plan4096_ab.txt; done
echo "[--chunk 3000 generic] $(timeout 300 $B --chunk 3000 2>/dev/null | python -c "$pickb")" | tee -a $O/plan4096_ab.txt
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5b_batch > $O/prof_batch.log 2>&1; echo "profile batch rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5b_chain --filter chain --chunk 8192 --fs 96000 > $O/prof_chain.log 2>&1; echo "profile chain rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5b_config4 --filter highcut --channels 8192 > $O/prof_config4.log 2>&1; echo "profile config4 rc=$?"
PROF_ONLY="1 2 4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5b_stream --mode stream --pipeline 1 > $O/prof_stream.log 2>&1; echo "profile stream rc=$?"
for t in batch chain config4 stream; do cp gpurun_out/prof_r5b_$t/summary.txt $O/${t}_summary.txt 2>/dev/null; find gpurun_out/prof_r5b_$t/trace -name '*kernel_stats.csv' -exec cp {} $O/${t}_kernel_stats.csv \; 2>/dev/null; done
head -4 $O/stream_summary.txt; grep -E "FETCH|WRITE" $O/*_summary.txt
