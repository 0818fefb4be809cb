#!/usr/bin/env python3
"""What happens after a test that FAILS while its ring steps ride a live session (the engine is never closed, pytest keeps the frame alive)?
Session 39's whole-suite run failed in such a test and aborted inside the NEXT test's engine creation.  This probe leaves engines with a
pipeline-owned session open - drained, mid-stream, and with their output tensors freed - and creates long-kernel engines behind them."""
import ctypes
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pyaudiodsptools_amd as adsp  # noqa: E402
from pyaudiodsptools_amd import FirEngine, FirStream, design  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
n, channels, steps = 512, 300, 24
fir = FirStream(design.lowcut_kernel(300, 44100, n), n)
rng = np.random.default_rng(5)
long_fir = FirStream(rng.standard_normal(40001) / 200.0, 20000, latency_chunks=2, lookahead=0)
kept = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    mode = it % 3
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    eng = FirEngine(fir, channels=channels, ring_slots=8, optimize_for="stream")
    assert eng.ring_set_pipeline("auto") == 3
    user = torch.cuda.Stream()
    sp = user.cuda_stream
    last = steps if mode != 1 else steps - 5
    for k in range(last):
        slot = eng.ring_acquire(sp)
        assert hip.hipMemcpyAsync(slot, x[k].data_ptr(), channels * n * 4, 3, sp) == 0
        eng.apply_ring(y[k], sp)
    if mode != 1:
        eng.ring_join(sp)
        user.synchronize()
    kept.append(eng)           # never closed
    if mode == 2:
        del x, y               # the session's output table still names these addresses
        gc.collect()
        torch.cuda.empty_cache()
    t0 = time.perf_counter()
    e2 = adsp.UpolsFirEngine(long_fir, channels=3)
    xx = torch.empty((2, 3, 20000), device="cuda").uniform_(-1, 1)
    yy = torch.empty_like(xx)
    e2.apply_device(xx, yy, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print(f"iteration {it} mode {mode}: long-kernel engine behind an open session ok ({(time.perf_counter() - t0) * 1e3:.0f} ms), finite={bool(torch.isfinite(yy).all())}", flush=True)
    e2.close()
print("done: no abort")
