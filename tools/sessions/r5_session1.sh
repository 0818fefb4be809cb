#!/bin/bash
# round-5 GPU session 1: the new code on the GPU for the first time - the uniformly partitioned engine, the counter-based generator,
# the bench line with configs 4 / 5 and the host oracle check - then the whole -m gpu suite, then the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $O/pytest_round5.log 2>&1
echo "pytest(round5) rc=$?"; tail -15 $O/pytest_round5.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "example4 or golden or config" > $O/pytest_parity.log 2>&1
echo "pytest(parity subset) rc=$?"; tail -5 $O/pytest_parity.log
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_round5.py > $O/pytest_all.log 2>&1
echo "pytest(all other gpu) rc=$?"; tail -8 $O/pytest_all.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench(default) rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5s1/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "runs", d["runs"]["value_msamples_s"], "parity", d["max_rel_err"], "oracle", d.get("oracle_check", {}).get("max_rel_err"))
    for k, c in d.get("configs", {}).items():
        print("  ", k, c.get("value"), c.get("roofline", {}).get("frac"), c.get("parity_max_rel_err"), c.get("oracle_max_rel_err"), c.get("error"))
    s = d["stream"]
    print("  stream", s.get("value"), s.get("us_per_step"), s.get("roofline_frac"), "one", s.get("one_stream", {}).get("us_per_step"), s.get("one_stream", {}).get("avg_kernel_us"))
    c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
    lv = c3.get("resident_live", {})
    print("  config3 step", c3.get("us_per_step"), c3.get("avg_kernel_us"), "live", lv.get("stream_producer", {}).get("us_per_step"), lv.get("host_producer", {}).get("us_per_step"))
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("load"))
except Exception as e:
    print("no line:", e)
PY
