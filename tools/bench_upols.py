#!/usr/bin/env python3
"""Long kernels (Example4.py:5, ModuleTestsGPU.py:35: chunk_size 88200 - CreateLowCutFilter 44 099 taps, CreateEQ3BandFFT 88 197): the
uniformly partitioned engine (adsp_upols_*: one forward transform per input block, frequency-domain delay line, one inverse per output
block) beside PartitionedFirEngine (one engine pass per kernel slice, summed in the output buffer).  Device-resident float32 batches,
one call per chunk (the reference's call pattern), wall clock over `--calls` calls with torch events; Msamples/s and the fraction of
the 8 TB/s roofline at 8 algorithmic bytes per sample.   usage: python tools/bench_upols.py [--channels 64 1024] [--calls 24]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[64, 1024])
    ap.add_argument("--calls", type=int, default=24)
    ap.add_argument("--only", default="", help="upols / partitioned: run just one engine kind (profiling)")
    ap.add_argument("--block", type=int, default=0, help="block size of the uniformly partitioned engine (default: the largest the delay allows)")
    a = ap.parse_args()
    import torch
    import pyaudiodsptools_amd as adsp
    from pyaudiodsptools_amd import design, synth
    n, fs = 88200, 44100
    dev = torch.device("cuda", 0)
    out = {}
    for name, taps in (("lowcut_44099_taps", design.lowcut_kernel(800, fs, n)), ("eq3_88197_taps", design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n))):
        fir = adsp.FirStream(taps, n)
        for C in a.channels:
            x = torch.empty((4, C, n), device=dev)
            synth.fill_device(x, 1234, 0, 0, C, n, 4, "f32", 1.0, 0, torch.cuda.current_stream().cuda_stream)
            y = torch.empty((C, n), device=dev)
            res = {}
            for kind, make in (("upols", lambda: adsp.UpolsFirEngine(fir, channels=C, block=a.block or None)), ("partitioned", lambda: adsp.PartitionedFirEngine(fir, channels=C))):
                if a.only and a.only != kind:
                    continue
                eng = make()
                s = torch.cuda.current_stream().cuda_stream
                for k in range(6):
                    eng.apply_device(x[k % 4], y, 1, s)
                torch.cuda.synchronize()
                t_pre = time.perf_counter()
                while time.perf_counter() - t_pre < 0.15:  # clock ramp
                    eng.apply_device(x[0], y, 1, s)
                    torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                runs = []
                for _ in range(3):
                    e0.record()
                    for k in range(a.calls):
                        eng.apply_device(x[k % 4], y, 1, s)
                    e1.record()
                    torch.cuda.synchronize()
                    runs.append(e0.elapsed_time(e1) * 1e-3 / a.calls)
                per = sorted(runs)[1]
                res[kind] = {"us_per_call": round(per * 1e6, 1), "msamples_s": round(C * n / per / 1e6, 1), "roofline_frac": round(8 * C * n / per / 8e12, 4),
                             "passes" if kind == "partitioned" else "partitions": len(eng.engines) if kind == "partitioned" else eng.partition.n_partitions}
                if kind == "upols":
                    res[kind]["delay_line_mib"] = round(eng.delay_line_bytes / 2 ** 20, 1)
                    res[kind]["block"] = eng.block
                eng.close()
                del eng
                torch.cuda.empty_cache()
            if "upols" in res and "partitioned" in res:
                res["speedup"] = round(res["partitioned"]["us_per_call"] / res["upols"]["us_per_call"], 2)
            out[f"{name}_{C}ch_x_{n}"] = res
            print(name, C, json.dumps(res), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
