#!/bin/bash
# round-4 GPU session 6: live step trace in steady state, different workgroups
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s6
for first in 3000; do for wg in 0 -1; do
ADSP_LIVE_TRACE=$first ADSP_LIVE_TRACE_WG=$wg timeout 120 python - 2>&1 <<'PY' | grep -v amdgpu.ids | grep "trace\|steps:" | tail -2
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
print("first", os.environ["ADSP_LIVE_TRACE"], "wg", os.environ["ADSP_LIVE_TRACE_WG"], "steps:", round((t1 - t0) / n * 1e6, 2), "us per step", flush=True)
PY
done; done > gpurun_out/r4s6/trace.txt 2>&1
cat gpurun_out/r4s6/trace.txt | cut -c1-330
