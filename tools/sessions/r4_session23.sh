#!/bin/bash
# round-4 GPU session 23: which workgroups pace a live session?  The per-step trace of one workgroup per run (the time it spends
# waiting for a publication is the slack it has over the slowest workgroup, whose arrival the producer's flow control follows)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s23
for wg in ${WGS:-0 1 2 3 4 5 6 7 8 64 512 1000 2048 3000 4080 4088 4089 4090 4091 4092 4093 4094 4095}; do
echo "== wg $wg"
ADSP_LIVE_TRACE=3000 ADSP_LIVE_TRACE_WG=$wg timeout 200 python - 2>&1 <<'PY' | grep -v amdgpu.ids | cut -c1-300
import os, sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
fir = bench.make_fir(a3)
dev = torch.device("cuda", 0)
C, N, ring, steps = 4096, 512, 256, 4096
geo = design.overlap_save_geometry(fir, 0, "stream")
eng = FirEngine(fir, channels=C, ring_slots=ring + geo.history_chunks)
scratch = torch.empty((C, N), device=dev)
s0 = torch.cuda.current_stream().cuda_stream
for _ in range(eng.ring_slots):
    eng.apply_device(torch.empty((C, N), device=dev).uniform_(-1, 1), scratch, 1, s0)
torch.cuda.synchronize()
out = torch.zeros((8, C, N), device=dev)
eng.live_configure(step_timeout_ms=10000.0, load_mode=2)
def session(n):
    eng.live_start(out, 8, n, None)
    time.sleep(0.002)
    t0 = time.perf_counter()
    eng.live_publish_run(n, None)
    eng.live_wait(n, 20000.0)
    t1 = time.perf_counter()
    assert eng.live_stop() == n
    return (t1 - t0) / n * 1e6
for _ in range(3): session(512)
runs = [round(session(steps), 3) for _ in range(2)]
sys.stderr.flush()
print("us/step", runs)
PY
done > gpurun_out/r4s23/wgs.txt 2>&1
grep "==\|trace\|us/step" gpurun_out/r4s23/wgs.txt | cut -c1-250
