#!/usr/bin/env python3
"""(GPU box) host cost of the publication calls and behaviour of resident ring launches under a lagging producer."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyaudiodsptools_amd import FirEngine, FirStream, design  # noqa: E402

n, fs, C = 512, 44100, 4096
fir = FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n), n)
per = 64
eng = FirEngine(fir, channels=C, ring_slots=per + 2)
prod = torch.cuda.Stream()
cons = torch.cuda.Stream()  # not the legacy default stream: it is implicitly ordered against `prod`
out = torch.empty((per, C, n), device="cuda")
x = torch.empty((C, n), device="cuda").uniform_(-1, 1)
for _ in range(eng.ring_slots):
    eng.apply_device(x, out[0], 1, cons.cuda_stream)
torch.cuda.synchronize()
eng.ring_reset_order()
# 1. host cost of begin + end
t0 = time.perf_counter()
for _ in range(per):
    eng.ring_produce_begin(prod)
    eng.ring_produce_end(prod)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"publish (first {per} calls): host {1e6 * (t1 - t0) / per:.2f} us per step, drained after {1e6 * (t2 - t0) / per:.2f} us per step")
eng.enable_kernel_timing(True)
eng.apply_ring_resident(out, per, cons.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(per):
    eng.ring_produce_begin(prod)
    eng.ring_produce_end(prod)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"publish (steady): host {1e6 * (t1 - t0) / per:.2f} us per step")
eng.apply_ring_resident(out, per, cons.cuda_stream)
torch.cuda.synchronize()
ms, k = eng.kernel_time()
print(f"resident launch, everything published before: {1e3 * ms / per:.2f} us per step, timed out {eng.ring_resident_timed_out()}")
# 2. consumer first, producer lagging (5 us host sleep granularity is not available: publish as fast as the host can)
for rep in range(3):
    t0 = time.perf_counter()
    eng.apply_ring_resident(out, per, cons.cuda_stream)
    for _ in range(per):
        eng.ring_produce_begin(prod)
        eng.ring_produce_end(prod)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms, k = eng.kernel_time()
    print(f"consumer first: wall {1e6 * (t1 - t0) / per:.2f} us per step, kernel {1e3 * ms / per:.2f} us per step, timed out {eng.ring_resident_timed_out()}")
