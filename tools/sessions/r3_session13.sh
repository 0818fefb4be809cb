#!/bin/bash
# round-3 GPU session 13: full suite on the final tree, then the profiles the bench line quotes (re-stamped afterwards)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s13; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest all: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
python bench.py --filter eq3 --chunk 512 --channels 4096 --no-cpu-baseline --no-latency --steps 8 --warmup 4 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["stream"]
print("config3 batch", d["value"], d["roofline"]["frac"], "| stream", s["us_per_step"], "graph", s["graph"]["us_per_step"], "resident", {k:v for k,v in s["resident"].items() if k!="note"})'
bash tools/sessions/r3_session9.sh > $O/profile.log 2>&1; tail -12 $O/profile.log
