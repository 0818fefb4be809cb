#!/bin/bash
# round-6 GPU session 22: small host calls of the standalone effects / delay lines / scans through the pinned, mapped host window (capi_common.hpp):
# the suites that call them, then the reference's timing harness again (session 21 = staging copies: 46 / 42 / 203 us per chunk)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s22
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_effects.py tests/test_gpu_callers.py tests/test_gpu_recursive.py tests/test_gpu_moduletests.py -q -m gpu -rf 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -15 | tee $O/pytest_subset.txt
for r in 1 2; do timeout 600 python examples/harness_timing.py > $O/harness_timing_$r.json 2>> $O/harness_timing.err; echo "harness rc=$?"; done
python - <<'PY'
import json
for r in (1, 2):
    d = json.load(open(f"gpurun_out/r6s22/harness_timing_{r}.json"))
    print({k: v["ms_per_chunk"] for k, v in d["ModuleTests.py"].items() if isinstance(v, dict)}, {k: v["ms_per_chunk"] for k, v in d["ModuleTestsGPU.py"].items() if isinstance(v, dict)})
PY
