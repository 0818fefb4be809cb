#!/bin/bash
# round-4 GPU session 11: the whole -m gpu suite, then the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s11
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r4s11/pytest_gpu.log 2>&1
echo "suite rc=$?"; tail -8 gpurun_out/r4s11/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r4s11/bench_default.json 2> gpurun_out/r4s11/bench_default.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s11/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["frac"], d["runs"]["value_msamples_s"], d["runs"]["shader_mhz"], "parity", d["max_rel_err"])
s = d["stream"]
print("stream", {k: s.get(k) for k in ("value", "us_per_step", "roofline_frac", "runs_us_per_step", "pipelined_error")})
print("one_stream", {k: s.get("one_stream", {}).get(k) for k in ("value", "us_per_step", "avg_kernel_us", "roofline_frac")}, "resident", s.get("resident", {}).get("us_per_step"))
c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
print("config3", {k: c3.get(k) for k in ("us_per_step", "avg_kernel_us")}, "pipelined", c3.get("pipelined", {}).get("us_per_step"), "resident", c3.get("resident", {}).get("us_per_step"))
print("live", json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in v if kk in ("us_per_step", "median", "p90", "min", "runs_us_per_step")}) for k, v in c3.get("resident_live", {}).items() if k != "note"}))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
tail -3 gpurun_out/r4s11/bench_default.err
