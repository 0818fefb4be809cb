#!/bin/bash
# round-3 GPU session 15: the final tree as the driver will see it - full -m gpu suite, smoke(), the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s15; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest all: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>$O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3s15/bench_driver_args.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype")})
print("roofline", d["roofline"])
print("config", d["config"]["workload"], "|", d["config"]["step"])
PY
