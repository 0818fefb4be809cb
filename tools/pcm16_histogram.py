#!/usr/bin/env python3
"""int16 front end on the GPU box: how often the float32 FFT engine's int16 output differs from the
reference's (y * 32767).astype(int16), and that the exact-mode engine does not; throughput of the exact mode.
Writes one JSON object (profiles/r2_pcm16_histogram.json is a copy)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fftfilter_oracle as o  # noqa: E402
from pyaudiodsptools_amd import ExactFirEngine, FirEngine, FirStream, design  # noqa: E402

out = {}
g = np.load(os.path.join(ROOT, "tests", "golden", "kat_example1.npz"))
pcm, want = g["pcm16_first8"], o.float_to_pcm16(g["out_first8"])
fir = FirStream(design.lowcut_kernel(800, 44100, 4096), 4096)


def hist(got, ref):
    d = got.astype(np.int32) - ref.astype(np.int32)
    return {"samples": int(d.size), "differ": int((d != 0).sum()), "fraction": float((d != 0).mean()),
            "minus_1": int((d == -1).sum()), "plus_1": int((d == 1).sum()), "beyond_1": int((np.abs(d) > 1).sum())}


for name, eng in (("fft_engine_float32", FirEngine(fir, channels=1, sample_format="s16", optimize_for="batch")),
                  ("exact_engine_float64", ExactFirEngine(fir, channels=1, sample_format="s16"))):
    y = eng.apply_host(pcm.reshape(8, 1, 4096)).reshape(-1)
    out[f"example1_first8_{name}_vs_reference"] = hist(y, want)
# full-scale random PCM, many channels: FFT engine and exact engine against the float64 truth through the reference's export
rng = np.random.default_rng(5)
C, steps, n = 64, 8, 4096
x = rng.integers(-30000, 30000, (steps, C, n), dtype=np.int16)
fe = FirEngine(fir, channels=C, sample_format="s16", optimize_for="batch").apply_host(x)
ee = ExactFirEngine(fir, channels=C, sample_format="s16").apply_host(x)
truth = np.stack([o.float_to_pcm16(o.direct_stream_convolution(fir.taps, o.pcm16_to_float(x[:, c].reshape(-1)), n).astype(np.float32))
                  for c in range(C)], axis=0).reshape(C, steps, n).transpose(1, 0, 2)
out["random_fullscale_fft_engine_vs_float64_truth"] = hist(fe, truth)
out["random_fullscale_exact_engine_vs_float64_truth"] = hist(ee, truth)
# exact-mode throughput: a bank of 64 Example1-sized files (264600 samples -> 65 chunks), device resident
import torch  # noqa: E402
C, steps = 64, 65
xd = torch.randint(-30000, 30000, (steps, C, n), dtype=torch.int16, device="cuda")
yd = torch.empty_like(xd)
eng = ExactFirEngine(fir, channels=C, sample_format="s16")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    eng.apply_device(xd, yd, steps, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.apply_device(xd, yd, steps, s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
out["exact_mode_throughput"] = {"msamples_per_s": round(C * steps * n / dt / 1e6, 1), "taps": len(fir.taps),
                                "gflops_f64": round(2 * len(fir.taps) * C * steps * n / dt / 1e9, 1),
                                "workload": f"{C} channels x {steps} chunks x {n} int16 samples, device resident"}
print(json.dumps(out, indent=1))
