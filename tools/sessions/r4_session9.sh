#!/bin/bash
# round-4 GPU session 9: live sessions with sharded arrival counters: 8 vs 16 points per thread
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s9
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu --timeout 300 -p no:cacheprovider -k "live" > gpurun_out/r4s9/pytest_live.log 2>&1
echo "live rc=$?" ; tail -3 gpurun_out/r4s9/pytest_live.log
for plan in 8 16; do
if [ $plan = 16 ]; then export ADSP_LIVE_PLAN16=1; else unset ADSP_LIVE_PLAN16; fi
ADSP_LIVE_TRACE=3000 ADSP_LIVE_TRACE_WG=7 timeout 300 python - 2>&1 <<'PY' | grep -v amdgpu.ids | cut -c1-400
import json, os, sys, torch
sys.path.insert(0, ".")
import bench
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
r = bench.live_figures(a3, bench.make_fir(a3), torch.device("cuda", 0), 8, 4096, 512, steps=4096, prewarm_ms=30.0)
sys.stderr.flush()
print("plan16" if os.environ.get("ADSP_LIVE_PLAN16") else "plan8", json.dumps({k: r[k] for k in ("stream_producer", "host_producer", "round_trip_us")}))
PY
done 2>&1 | grep "plan\|trace" | tail -8
