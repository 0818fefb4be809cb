#!/usr/bin/env python3
"""Run config 3 as live sessions (adsp_live_*) under rocprofv3 --kernel-trace: the persistent launch shows up as ONE dispatch of
adsp::fftconv_live_kernel per session, whose duration / steps is the per-step time the bench line quotes (tools/sessions/r4_session14.sh)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyaudiodsptools_amd import FirEngine, design  # noqa: E402

a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
fir = bench.make_fir(a3)
dev = torch.device("cuda", 0)
C, N, ring, steps = 4096, 512, 256, 8192
geo = design.overlap_save_geometry(fir, 0, "stream")
eng = FirEngine(fir, channels=C, ring_slots=ring + geo.history_chunks)
scratch = torch.empty((C, N), device=dev)
s0 = torch.cuda.current_stream().cuda_stream
for _ in range(eng.ring_slots):
    eng.apply_device(torch.empty((C, N), device=dev).uniform_(-1, 1), scratch, 1, s0)
torch.cuda.synchronize()
out = torch.empty((8, C, N), device=dev)
prod = torch.cuda.Stream()
res = []
for how in os.environ.get("LIVE_PRODUCERS", "host,stream,host,stream").split(","):  # PMC passes: "host,host" (rocprofv3 serialises kernels: no publishing kernel can run beside the session)
    eng.live_start(out, 8, steps, None)
    time.sleep(0.002)
    t0 = time.perf_counter()
    eng.live_publish_run(steps, prod if how == "stream" else None)
    eng.live_wait(steps, 30000.0)
    t1 = time.perf_counter()
    assert eng.live_stop() == steps
    res.append({"producer": how, "steps": steps, "wall_us_per_step": round((t1 - t0) / steps * 1e6, 3)})
print(json.dumps(res))
