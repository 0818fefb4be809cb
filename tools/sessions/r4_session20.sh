#!/bin/bash
# round-4 GPU session 20: what bounds a live-session step?  Ablations of the live kernel (results then wrong): pass twiddles as
# constants (1), pair tables as constants (2), no output stores (4), + no LDS exchange (7x)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s20
for v in ${VARIANTS:-live_glob live_abl1 live_abl3 live_abl4 live_abl7 live_abl7x live_glob}; do
ADSP_LIB=$PWD/abl/$v.so ADSP_LIVE_TRACE=3000 ADSP_LIVE_TRACE_WG=7 timeout 200 python - 2>&1 <<'PY' | grep -v amdgpu.ids | cut -c1-600
import json, os, sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", os.environ.get("CH", "4096")])
fir = bench.make_fir(a3)
dev = torch.device("cuda", 0)
C, N, ring, steps = int(os.environ.get("CH", "4096")), 512, 256, 4096
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
for _ in range(4): session(512)
runs = sorted(round(session(steps), 3) for _ in range(3))
sys.stderr.flush()
print(os.path.basename(os.environ["ADSP_LIB"]), "host producer us/step", runs)
PY
done > gpurun_out/r4s20/variants.txt 2>&1
grep "us/step\|trace" gpurun_out/r4s20/variants.txt | awk '/trace/{n++; if(n%7==6)print} /us.step/{print; n=0}' | cut -c1-260
