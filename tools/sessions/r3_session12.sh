#!/bin/bash
# round-3 GPU session 12: what the driver runs - smoke(), the default bench line, the driver's arguments, the chain line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s12; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>$O/bench_driver_args.err
python bench.py --filter chain --chunk 8192 --fs 96000 --no-latency --no-cpu-baseline > $O/bench_chain.json 2>$O/bench_chain.err
python bench.py --filter highcut --channels 8192 --no-latency --no-cpu-baseline > $O/bench_config4.json 2>$O/bench_config4.err
python - <<'PY'
import json
for name in ("bench_default", "bench_driver_args", "bench_chain", "bench_config4"):
    d = json.loads(open(f"gpurun_out/r3s12/{name}.json").read().strip().splitlines()[-1])
    s = d.get("stream", {})
    print(name, "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "ms/step", d["ms_per_step"],
          "| stream", s.get("value"), s.get("roofline_frac"), "two", s.get("two_streams", {}).get("roofline_frac"), "resident", {k: v for k, v in s.get("resident", {}).items() if k != "note"})
    if "latency" in d:
        l = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
        print("   config3:", l.get("us_per_step"), "graph", l.get("graph", {}).get("us_per_step"), "resident", {k: v for k, v in l.get("resident", {}).items() if k != "note"}, "numpy api", d["latency"].get("numpy_api_apply_us_per_call"))
    if "cpu_baseline" in d:
        print("   cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
