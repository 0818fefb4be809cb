#!/bin/bash
# round-4 GPU session 5: shader clock while a live session runs / while per-step launches run
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s5
timeout 300 python - > gpurun_out/r4s5/clock.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
from pyaudiodsptools_amd.engine import ClockProbe
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
ps = torch.cuda.Stream()
p = ClockProbe(0, 2000.0, ps.cuda_stream); print("idle clock MHz", round(p.read(), 1))
for rep in range(3):
    n = 8192
    eng.live_start(out, 8, n, None)
    t0 = time.perf_counter()
    probe = ClockProbe(0, 20000.0, ps.cuda_stream)
    eng.live_publish_run(n, None)
    eng.live_wait(n, 20000.0)
    t1 = time.perf_counter()
    eng.live_stop()
    print("live session", n, "steps:", round((t1 - t0) / n * 1e6, 2), "us per step, shader clock MHz", round(probe.read(), 1), flush=True)
for rep in range(2):
    probe = ClockProbe(0, 20000.0, ps.cuda_stream)
    t0 = time.perf_counter()
    for i in range(8192):
        eng.apply_ring(out[i % 8], s0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("per-step launches:", round((t1 - t0) / 8192 * 1e6, 2), "us per step, shader clock MHz", round(probe.read(), 1), flush=True)
PY
grep -v amdgpu.ids gpurun_out/r4s5/clock.txt
