#!/bin/bash
# round-3 GPU session 6: full -m gpu suite (32-channel oracle checks, full Example1), default bench line, resident probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s6; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 ) > $O/pytest.log 2>&1
echo "pytest all: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head; grep -E "^[0-9.]+s " $O/pytest.log | head -8
timeout 300 python tools/resident_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3s6/bench_default.json").read().strip().splitlines()[-1])
s, l = d["stream"], d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("stream", s["value"], s["roofline_frac"], "graph", s["graph"]["us_per_step"], "two", s["two_streams"]["us_per_step"], s["two_streams"]["roofline_frac"], "resident", s.get("resident"))
print("config3", {k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in v if kk != "note"}) for k, v in l.items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
