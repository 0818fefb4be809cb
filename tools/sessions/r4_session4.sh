#!/bin/bash
# round-4 GPU session 4: where a live step's time goes (per-phase trace), pipelined-stream variants, live tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s4
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu --timeout 300 -p no:cacheprovider -k "live or pipelined" > gpurun_out/r4s4/pytest_live.log 2>&1
echo "live rc=$?" ; tail -5 gpurun_out/r4s4/pytest_live.log
ADSP_LIVE_TRACE=1 timeout 300 python - > gpurun_out/r4s4/trace.txt 2>&1 <<'PY'
import json, sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
dev = torch.device("cuda", 0)
fir = bench.make_fir(a3)
for ch in (4096, 512):
    for mode in (2, 0):
        print("== channels", ch, "load_mode", mode, flush=True)
        r = bench.live_figures(a3, fir, dev, 8, ch, 512, steps=2048, load_mode=mode, prewarm_ms=20.0)
        print(json.dumps({k: r[k] for k in ("stream_producer", "host_producer", "round_trip_us")}), flush=True)
        sys.stderr.flush()
PY
grep -v amdgpu.ids gpurun_out/r4s4/trace.txt | cut -c1-400
timeout 300 python - > gpurun_out/r4s4/pipe.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
args = bench.parse([])
fir = bench.make_fir(args)
dev = torch.device("cuda", 0)
C, N = 4096, 4096
geo = design.overlap_save_geometry(fir, 0, "stream")
def fill(eng):
    scratch = torch.empty((C, N), device=dev)
    for _ in range(eng.ring_slots):
        eng.apply_device(torch.empty((C, N), device=dev).uniform_(-1, 1), scratch, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
outs = [torch.empty((C, N), device=dev) for _ in range(4)]
def timeit(fn, steps=1024):
    fn(256); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(steps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6
for slots in (4, 5, 8):
    eng = FirEngine(fir, channels=C, ring_slots=slots); fill(eng)
    null = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    def one(k):
        for i in range(k): eng.apply_ring(outs[i % 4], null)
    print("slots", slots, "one stream (NULL)", round(timeit(one), 2))
    s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
    def two(k):
        for i in range(k):
            st = s2[i % 2].cuda_stream
            eng.ring_acquire(st); eng.apply_ring(outs[i % 4], st)
    eng.ring_reset_order()
    print("slots", slots, "caller-managed two streams", round(timeit(two), 2))
    eng.ring_reset_order()
    eng.ring_set_pipeline(2)
    def pipe_null(k):
        for i in range(k):
            eng.ring_acquire(null); eng.apply_ring(outs[i % 4], null)
        eng.ring_join(null)
    print("slots", slots, "library pipeline, user = NULL stream", round(timeit(pipe_null), 2))
    def pipe_side(k):
        for i in range(k):
            eng.ring_acquire(side.cuda_stream); eng.apply_ring(outs[i % 4], side.cuda_stream)
        eng.ring_join(side.cuda_stream)
    print("slots", slots, "library pipeline, user = side stream", round(timeit(pipe_side), 2))
    def pipe_noacq(k):
        for i in range(k):
            eng.apply_ring(outs[i % 4], side.cuda_stream)
        eng.ring_join(side.cuda_stream)
    print("slots", slots, "library pipeline, no acquire", round(timeit(pipe_noacq), 2), flush=True)
    eng.close()
PY
grep -v amdgpu.ids gpurun_out/r4s4/pipe.txt
