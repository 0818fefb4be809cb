#!/bin/bash
# round-4 GPU session 10: is the live session paced by its flow control?  ring length sweep, relay iteration count
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
ADSP_DEBUG=1 timeout 300 python - 2>&1 <<'PY' | grep -v amdgpu.ids | grep "ring\|relay" | sed 's/progress:.*//' | cut -c1-300
import json, os, sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
fir = bench.make_fir(a3)
dev = torch.device("cuda", 0)
C, N = 4096, 512
geo = design.overlap_save_geometry(fir, 0, "stream")
for ring in (32, 256, 1024):
    eng = FirEngine(fir, channels=C, ring_slots=ring + geo.history_chunks)
    scratch = torch.empty((C, N), device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    for _ in range(eng.ring_slots):
        eng.apply_device(torch.empty((C, N), device=dev).uniform_(-1, 1), scratch, 1, s0)
    torch.cuda.synchronize()
    out = torch.empty((8, C, N), device=dev)
    for rep in range(2):
        n = 8192
        eng.live_start(out, 8, n, None)
        time.sleep(0.002)
        t0 = time.perf_counter()
        eng.live_publish_run(n, None)
        eng.live_wait(n, 20000.0)
        t1 = time.perf_counter()
        eng.live_stop()
        sys.stderr.flush()
        print("ring", ring, "us per step", round((t1 - t0) / n * 1e6, 2), flush=True)
    eng.close()
PY
