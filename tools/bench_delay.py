#!/usr/bin/env python3
"""Throughput of the tapped delay line kernel on the headline batch shape; run on the GPU box.

    python tools/bench_delay.py [--channels 4096] [--chunk 4096] [--steps 32] [--ms 500] [--loops 2]

Prints one JSON line: Msamples/s, kernel us per launch (torch events on the launch stream) and the HBM roofline fraction
for the algorithmic traffic of (taps + 2) * 4 bytes per sample (input, one read per tap, output)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyaudiodsptools_amd as adsp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--channels", type=int, default=4096)
ap.add_argument("--chunk", type=int, default=4096)
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--ms", type=float, default=500)
ap.add_argument("--loops", type=int, default=2)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()

adsp.config.initialize(44100, a.chunk)
d = adsp.CreateDelay(a.ms, a.loops, channels=a.channels)
x = torch.rand((a.steps, a.channels, a.chunk), device="cuda") * 2 - 1
y = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
import time  # noqa: E402
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.3:  # clock ramp: sustained regime before timing
    d.line.apply_device(x, y, a.steps, s)
    torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.reps):
    d.line.apply_device(x, y, a.steps, s)
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / a.reps
samples = a.steps * a.channels * a.chunk
bytes_alg = (a.loops + 2) * 4 * samples
print(json.dumps({"kernel": "delay_kernel", "workload": f"CreateDelay({a.ms} ms, {a.loops} loops), {a.channels} ch x {a.chunk} x {a.steps} steps/launch",
                  "Msamples_per_s": round(samples / ms / 1e3, 1), "us_per_launch_incl_ring_update": round(ms * 1e3, 1),
                  "roofline": {"bound": "hbm", "achieved_GBps": round(bytes_alg / ms / 1e6, 1), "peak": 8000.0,
                               "frac": round(bytes_alg / ms / 1e6 / 8000.0, 4), "bytes_per_sample": (a.loops + 2) * 4}}))
