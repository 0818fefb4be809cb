#!/usr/bin/env python3
"""Throughput of the standalone elementwise kernels (adsp_effect_device, adsp_mix_device) on 1 GiB; run on the GPU box."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyaudiodsptools_amd as adsp  # noqa: E402
from pyaudiodsptools_amd import _capi  # noqa: E402

n = 1 << 28
x = torch.rand(n, device="cuda") * 3 - 1.5
lib = _capi.load()


def timed(f, reps=40):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:  # clock ramp
        f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


adsp.config.initialize(44100, 4096)
for name, eff in (("volume", adsp.CreateVolumeChange(-3.0)), ("soft_clipper", adsp.CreateSoftClipper()),
                  ("saturator", adsp.CreateSaturator()), ("tremolo", adsp.CreateTremolo())):
    ms = timed(lambda: eff.apply(x))
    print(json.dumps({"kernel": "adsp_pointwise_kernel<" + name + ">", "Msamples_per_s": round(n / ms / 1e3, 1),
                      "hbm_GBps": round(8 * n / ms / 1e6, 1), "roofline_frac": round(8 * n / ms / 1e6 / 8000, 4)}))
ins = [torch.rand(n // 4, device="cuda") for _ in range(3)]
out = torch.empty(n // 4, device="cuda")
ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ins])
ms = timed(lambda: _capi.check(lib.adsp_mix_device(0, ptrs, 3, 1, ctypes.c_void_p(out.data_ptr()), n // 4, None)))
print(json.dumps({"kernel": "adsp_mix_kernel<3 inputs>", "Msamples_out_per_s": round(n / 4 / ms / 1e3, 1),
                  "hbm_GBps": round(16 * (n // 4) / ms / 1e6, 1), "roofline_frac": round(16 * (n // 4) / ms / 1e6 / 8000, 4)}))
