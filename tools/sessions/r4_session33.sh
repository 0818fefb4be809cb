#!/bin/bash
# round-4 GPU session 33: the default bench line on the final tree (8-byte accesses in live sessions)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s33
timeout 900 python bench.py > gpurun_out/r4s33/bench_default.json 2> gpurun_out/r4s33/bench_default.err
echo "bench(default) rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s33/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "runs", d["runs"]["value_msamples_s"], "parity", d["max_rel_err"])
s = d["stream"]
print("  stream", s.get("value"), s.get("us_per_step"), s.get("roofline_frac"), s.get("runs_us_per_step"), "one", s.get("one_stream", {}).get("us_per_step"), s.get("one_stream", {}).get("avg_kernel_us"), "graph", s.get("one_stream", {}).get("graph", {}).get("us_per_step"), "resident", s.get("resident", {}).get("us_per_step"), s.get("pipelined_error"))
c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
lv = c3.get("resident_live", {})
print("  config3 step", c3.get("us_per_step"), "pipelined", c3.get("pipelined"), "resident", c3.get("resident", {}).get("us_per_step"), "live", lv.get("stream_producer", {}).get("us_per_step"), lv.get("host_producer", {}).get("us_per_step"), "rt", lv.get("round_trip_us", {}).get("median"))
PY
