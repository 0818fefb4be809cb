#!/bin/bash
# round-2 GPU session 12: full suite on the final tree, build() + smoke(), default bench line, all BASELINE shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s12; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k:d["roofline"][k] for k in ("achieved","frac","traffic","avg_launch_us")})
print("stream", {k:d["stream"].get(k) for k in ("value","us_per_step","avg_kernel_us","roofline_frac")}, "graph", d["stream"].get("graph",{}).get("us_per_step"))
print("latency", d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]["us_per_step"], d["latency"]["config3_eq3_2048_stereo_pairs_x_512"].get("graph",{}).get("us_per_step"), d["latency"]["numpy_api_apply_us_per_call"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["variants_msamples_s"])
PY
bash tools/bench_shapes.sh > $O/shapes.txt 2>&1; cat $O/shapes.txt
