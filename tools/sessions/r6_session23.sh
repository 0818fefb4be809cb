#!/bin/bash
# round-6 GPU session 23: the whole GPU suite on the library with the pinned host window (effects / mix / delay / scan host calls), smoke, the default line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s23
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED" $O/pytest_all.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py 2>/dev/null > $O/bench_default.json; python -c 'import json; d=json.load(open("gpurun_out/r6s23/bench_default.json")); print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], {k: v["roofline"]["frac"] for k, v in d["configs"].items()})'
timeout 600 python examples/harness_timing.py > $O/harness_timing.json 2> $O/harness_timing.err; echo "harness rc=$?"
