#!/bin/bash
# round-4 GPU session 16: resident launches with a LIVE per-step producer, consumer on a high-priority stream (its own pool of
# hardware queues): was the round-3 limitation (the producer's commands stuck behind the half-dispatched grid) queue sharing?
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s16
timeout 300 python - > gpurun_out/r4s16/resident_live.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
from pyaudiodsptools_amd import FirEngine, design
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
fir = bench.make_fir(a3)
dev = torch.device("cuda", 0)
C, N, n = 4096, 512, 128
geo = design.overlap_save_geometry(fir, 0, "stream")
for prio, label in ((-1, "consumer on a HIGH-priority stream"), (0, "consumer on a normal stream")):
    eng = FirEngine(fir, channels=C, ring_slots=2 * n + geo.history_chunks)
    scratch = torch.empty((C, N), device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    for _ in range(eng.ring_slots):
        eng.apply_device(torch.empty((C, N), device=dev).uniform_(-1, 1), scratch, 1, s0)
    torch.cuda.synchronize()
    eng.ring_reset_order()
    eng.ring_resident_timeout(100.0)
    out = torch.empty((n, C, N), device=dev)
    prod = torch.cuda.Stream()
    cons = torch.cuda.Stream(priority=prio)
    def run(launches, lead):
        # `lead` steps are published before the consumer launch, the others after it, one publication per step
        for L in range(launches):
            for k in range(lead):
                eng.ring_produce_begin(prod); eng.ring_produce_end(prod)
            eng.apply_ring_resident(out, n, cons)
            for k in range(n - lead):
                eng.ring_produce_begin(prod); eng.ring_produce_end(prod)
    for lead in (n, 8, 0):
        try:
            run(2, lead); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(12, lead); torch.cuda.synchronize(); t1 = time.perf_counter()
            to = eng.ring_resident_timed_out()
            print(label, "| lead", lead, "| us per step", round((t1 - t0) / (12 * n) * 1e6, 2), "| timed out", to, flush=True)
        except Exception as exc:
            print(label, "| lead", lead, "| error", str(exc)[:120], flush=True)
            torch.cuda.synchronize()
        eng.ring_reset_order()
    eng.close()
PY
grep -v amdgpu.ids gpurun_out/r4s16/resident_live.txt
