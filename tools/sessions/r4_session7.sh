#!/bin/bash
# round-4 GPU session 7: is it the relay that slows its SIMD mates?  (relay off, device-side publication, ring >= steps)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s7
for off in 0 1; do for wg in 0 5 -1; do
if [ $off = 1 ]; then export ADSP_LIVE_RELAY_OFF=1; else unset ADSP_LIVE_RELAY_OFF; fi
ADSP_LIVE_TRACE=100 ADSP_LIVE_TRACE_WG=$wg timeout 120 python - 2>&1 <<'PY' | grep -v amdgpu.ids | grep "trace\|relay" | tail -2
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
n = 250
prod = torch.cuda.Stream()
eng.live_configure(step_timeout_ms=300.0)
eng.live_start(out, 8, n, None)
t0 = time.perf_counter()
for k in range(n):
    eng.live_slot()
eng.live_publish(prod)       # ONE device-side publication of all 250 steps
if os.environ.get("ADSP_LIVE_RELAY_OFF"):
    time.sleep(0.05)
else:
    eng.live_wait(n, 5000.0)
t1 = time.perf_counter()
try:
    eng.live_stop()
except Exception as exc:
    print("stop:", str(exc)[:80])
print("relay off" if os.environ.get("ADSP_LIVE_RELAY_OFF") else "relay on", "wg", os.environ["ADSP_LIVE_TRACE_WG"], flush=True)
PY
done; done > gpurun_out/r4s7/trace.txt 2>&1
cut -c1-330 gpurun_out/r4s7/trace.txt
