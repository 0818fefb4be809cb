#!/bin/bash
# round-4 GPU session 8: live sessions with the scalar-polling relay and arrival counters: tests, trace, figures
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s8
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu --timeout 300 -p no:cacheprovider -k "live" > gpurun_out/r4s8/pytest_live.log 2>&1
echo "live rc=$?" ; tail -5 gpurun_out/r4s8/pytest_live.log
for wg in 0 -1; do
ADSP_LIVE_TRACE=3000 ADSP_LIVE_TRACE_WG=$wg timeout 120 python - 2>&1 <<'PY' | grep -v amdgpu.ids | grep "trace\|steps:" | tail -2 | cut -c1-330
import os, sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
fir = bench.make_fir(a3)
dev = torch.device("cuda", 0)
C, N = 4096, 512
geo = design.overlap_save_geometry(fir, 0, "stream")
eng = FirEngine(fir, channels=C, ring_slots=256 + geo.history_chunks)
scratch = torch.empty((C, N), device=dev)
s0 = torch.cuda.current_stream().cuda_stream
for _ in range(eng.ring_slots):
    eng.apply_device(torch.empty((C, N), device=dev).uniform_(-1, 1), scratch, 1, s0)
torch.cuda.synchronize()
out = torch.empty((8, C, N), device=dev)
n = 8192
eng.live_start(out, 8, n, None)
t0 = time.perf_counter()
eng.live_publish_run(n, None)
eng.live_wait(n, 20000.0)
t1 = time.perf_counter()
eng.live_stop()
print("wg", os.environ["ADSP_LIVE_TRACE_WG"], "steps:", round((t1 - t0) / n * 1e6, 2), "us per step", flush=True)
PY
done
timeout 300 python - 2>&1 <<'PY' | grep -v amdgpu.ids
import json, sys, torch
sys.path.insert(0, ".")
import bench
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
r = bench.live_figures(a3, bench.make_fir(a3), torch.device("cuda", 0), 8, 4096, 512, steps=4096)
print(json.dumps({k: r[k] for k in ("stream_producer", "host_producer", "round_trip_us")}))
PY
