#!/usr/bin/env python3
"""Blocks of 32768 (a tuning build: abl/b32k.so, Plan<32768, 32 points, radices 32 x 8 x 8 x 16> in 1024 threads) against blocks of 16384: same samples to 1e-5,
and the time per call at chunk 88200.   usage: ADSP_LIB=abl/b32k.so python tools/probe_block_32768.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pyaudiodsptools_amd as adsp  # noqa: E402
from pyaudiodsptools_amd import design  # noqa: E402

n, fs = 88200, 44100
out = {"block_sizes": list(adsp.UpolsFirEngine.block_sizes())}
for name, taps in (("lc", design.lowcut_kernel(800, fs, n)), ("eq", design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n))):
    fir = adsp.FirStream(taps, n)
    x = torch.empty((5, 3, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(3))
    ys = {}
    for b in (16384, 32768):
        eng = adsp.UpolsFirEngine(fir, channels=3, block=b)
        y = torch.empty_like(x)
        s = torch.cuda.current_stream().cuda_stream
        for k in range(5):
            eng.apply_device(x[k], y[k], 1, s)
        torch.cuda.synchronize()
        ys[b] = y.clone()
        eng.close()
    scale = float(ys[16384].abs().max())
    out[name + "_max_rel_diff"] = float((ys[16384] - ys[32768]).abs().max()) / scale
    for C in (256, 1024):
        xx = torch.empty((4, C, n), device="cuda").uniform_(-1, 1)
        yy = torch.empty((C, n), device="cuda")
        for b in (16384, 32768):
            eng = adsp.UpolsFirEngine(fir, channels=C, block=b)
            s = torch.cuda.current_stream().cuda_stream
            for k in range(12):
                eng.apply_device(xx[k % 4], yy, 1, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            runs = []
            for _ in range(3):
                e0.record()
                for k in range(24):
                    eng.apply_device(xx[k % 4], yy, 1, s)
                e1.record()
                torch.cuda.synchronize()
                runs.append(e0.elapsed_time(e1) * 1e3 / 24)
            out[f"{name}_{C}ch_block{b}_us"] = round(sorted(runs)[1], 1)
            eng.close()
print(json.dumps(out))
