#!/usr/bin/env python3
"""Throughput of the per-channel scan kernels (IIR 3-band EQ cascade, compressor, gate) on the headline batch shape; GPU box."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyaudiodsptools_amd as adsp  # noqa: E402

C, N, STEPS = 4096, 4096, 8
adsp.config.initialize(44100, N)
x = (torch.rand((STEPS, C, N), device="cuda") * 2 - 1) * 0.3
y = torch.empty_like(x)
eq = adsp.CreateEQ3Band(100, 2, 700, -4, 8000, 5, channels=C)
cp = adsp.CreateCompressor(channels=C)
gt = adsp.CreateGate(channels=C)
for name, eng in (("iir_eq3_cascade", eq.cascade), ("iir_one_band", eq._low), ("compressor", cp.engine), ("gate", gt.engine)):
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.3:  # clock ramp
        eng.apply_device(x, y, STEPS)
        torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        eng.apply_device(x, y, STEPS, torch.cuda.current_stream().cuda_stream)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 3
    samples = STEPS * C * N
    print(json.dumps({"kernel": "scan_kernel<" + name + ">", "workload": f"{C} ch x {N} x {STEPS} steps/launch",
                      "Msamples_per_s": round(samples / ms / 1e3, 1), "ms_per_launch": round(ms, 2),
                      "hbm_GBps_algorithmic": round(8 * samples / ms / 1e6, 1), "roofline_frac": round(8 * samples / ms / 1e6 / 8000, 4)}))
