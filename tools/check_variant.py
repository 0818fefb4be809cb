#!/usr/bin/env python3
"""Correctness of a tuning plan variant (ADSP_PLAN_VARIANT=<i>) on the GPU box: cut filter and EQ at chunk N against the
float64 direct convolution.  usage: ADSP_PLAN_VARIANT=4 python tools/check_variant.py 512 [fft_mult]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fftfilter_oracle as o  # noqa: E402
from pyaudiodsptools_amd import FirEngine, FirStream, design  # noqa: E402

n = int(sys.argv[1])
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fs, C, steps = 44100, 5, 6
x = np.random.default_rng(n).uniform(-1, 1, (steps, C, n)).astype(np.float32)
for name, fir in (("lowcut", FirStream(design.lowcut_kernel(300, fs, n), n)),
                  ("eq3", FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n), n))):
    for opt in ("stream", "batch"):
        eng = FirEngine(fir, channels=C, fft_mult=mult, optimize_for=opt)
        ys = np.stack([eng.apply_host(x[k]) for k in range(steps)])
        eng.reset()
        yb = eng.apply_host(x)
        worst = 0.0
        for c in range(C):
            t = o.direct_stream_convolution(fir.taps, x[:, c].reshape(-1), n, fir.latency_chunks, fir.lookahead)
            worst = max(worst, np.abs(ys[:, c].reshape(-1) - t).max() / np.abs(t).max(), np.abs(yb[:, c].reshape(-1) - t).max() / np.abs(t).max())
        print(f"variant {os.environ.get('ADSP_PLAN_VARIANT')} N={n} {name} {opt} F={eng.geometry.fft_size} plan={eng.plan}: max rel err {worst:.2e} {'OK' if worst < 1e-5 else 'FAIL'}")
