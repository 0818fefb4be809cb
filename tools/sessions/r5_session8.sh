#!/bin/bash
# round-5 GPU session 8: the final plan table (M = 4096: XL for F = 2N, two waves for F = 4N): the whole -m gpu suite, the default bench
# line, the PMC passes for the traffic stamps, and a generic-geometry bench shape.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s8
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo "bench(default) rc=$?"; tail -3 $O/bench_time.txt
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5s8/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "parity", d["max_rel_err"], "oracle", d["oracle_check"]["max_rel_err"])
    for k, c in d.get("configs", {}).items(): print("  ", k, c.get("value"), c["roofline"]["frac"], c["roofline"]["traffic"])
    s = d["stream"]
    print("  stream", s.get("value"), s.get("us_per_step"), s.get("roofline_frac"), "one", s["one_stream"]["us_per_step"], s["one_stream"]["avg_kernel_us"], "resident", s.get("resident", {}).get("us_per_step"), "live_pipeline", str(s.get("live_pipeline"))[:80])
    c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
    lv = c3.get("resident_live", {})
    print("  config3", c3.get("us_per_step"), c3.get("launch_per_step", {}).get("us_per_step"), "live", lv.get("stream_producer", {}).get("us_per_step"), lv.get("host_producer", {}).get("us_per_step"), lv.get("round_trip_us", {}).get("median"))
    print("  numpy_api", d["latency"]["numpy_api_apply_us_per_call"], json.dumps(d["latency"]["numpy_api"]["apply_host_1gib"])[:300], json.dumps(d["latency"]["numpy_api"].get("wavbank_process"))[:300])
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("load"))
except Exception as e:
    print("no line:", e)
PY
timeout 300 python bench.py --chunk 3000 --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 4 --warmup 1 --runs 3 > $O/bench_chunk3000.json 2> $O/bench_chunk3000.err
echo "bench(chunk 3000) rc=$?"; tail -4 $O/bench_chunk3000.err; cut -c1-300 $O/bench_chunk3000.json
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5c_batch > $O/prof_batch.log 2>&1; echo "profile batch rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5c_chain --filter chain --chunk 8192 --fs 96000 > $O/prof_chain.log 2>&1; echo "profile chain rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5c_config4 --filter highcut --channels 8192 > $O/prof_config4.log 2>&1; echo "profile config4 rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5c_stream --mode stream --pipeline 1 > $O/prof_stream.log 2>&1; echo "profile stream rc=$?"
for t in batch chain config4 stream; do cp gpurun_out/prof_r5c_$t/summary.txt $O/${t}_summary.txt 2>/dev/null; find gpurun_out/prof_r5c_$t/trace -name '*kernel_stats.csv' -exec cp {} $O/${t}_kernel_stats.csv \; 2>/dev/null; done
grep -E "fftconv|FETCH|WRITE" $O/*_summary.txt | cut -c1-200
