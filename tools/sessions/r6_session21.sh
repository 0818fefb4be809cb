#!/bin/bash
# round-6 GPU session 21: the reference's timing harness through this package (examples/harness_timing.py) and a soak of the whole GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s21
mkdir -p $O
timeout 600 python examples/harness_timing.py > $O/harness_timing.json 2> $O/harness_timing.err; echo "harness rc=$?"; cat $O/harness_timing.json | tr -d '\n ' | cut -c1-1500; echo
timeout 600 python examples/harness_timing.py > $O/harness_timing_2.json 2>> $O/harness_timing.err; echo "harness(2) rc=$?"
for r in 1 2 3; do timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1 | tee -a $O/pytest_soak.txt; done
timeout 300 python bench.py 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_source"))' | tee $O/bench_traffic_check.txt
