#!/bin/bash
# round-4 GPU session 28: final tree after the live-session rework - the whole -m gpu suite, smoke(), the bench line with the driver's arguments and by default
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s28
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r4s28/pytest_gpu.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r4s28/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4s28/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4s28/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4s28/bench_driver_args.json 2> gpurun_out/r4s28/bench_driver_args.err
echo "bench(driver args) rc=$?"
timeout 900 python bench.py > gpurun_out/r4s28/bench_default.json 2> gpurun_out/r4s28/bench_default.err
echo "bench(default) rc=$?"
python - <<'PY'
import json
for f in ("bench_driver_args", "bench_default"):
    d = json.loads(open(f"gpurun_out/r4s28/{f}.json").read().strip().splitlines()[-1])
    print(f, "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "runs", d["runs"]["value_msamples_s"], "MHz", d["runs"]["shader_mhz"], "parity", d["max_rel_err"])
    s = d["stream"]
    print("  stream", s.get("value"), s.get("us_per_step"), s.get("roofline_frac"), "one", s.get("one_stream", {}).get("us_per_step"), "resident", s.get("resident", {}).get("us_per_step"))
    c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
    lv = c3.get("resident_live", {})
    print("  config3 step", c3.get("us_per_step"), "resident", c3.get("resident", {}).get("us_per_step"), "live", lv.get("stream_producer", {}).get("us_per_step"), lv.get("host_producer", {}).get("us_per_step"), "rt", lv.get("round_trip_us", {}).get("median"), lv.get("error"))
    print("  numpy api", d["latency"].get("numpy_api_apply_us_per_call"), "cpu", d["cpu_baseline"]["value"])
PY
