#!/usr/bin/env python3
"""The reference's timing harness against this package: ModuleTests.py times ten loops `chunks[i] = dev.apply(chunks[i])` over a chunked
1 kHz sine at 44100 Hz / 512 samples and prints `Total Time` / `Time per Chunk` for each device (ModuleTests.py:95-214); ModuleTestsGPU.py
does the same for the three FFT devices at chunk 88200 on a device array (:78-110).  This script runs both sections through
`compat.install()` - the reference's import lines, one chunk per call through the numpy (or device-tensor) API, which is the latency a
script written against the reference sees; the batched figures are bench.py's.

    python examples/harness_timing.py [--seconds 60]
"""
import argparse
import copy
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyaudiodsptools_amd.compat  # noqa: E402

pyaudiodsptools_amd.compat.install()
import pyAudioDspTools  # noqa: E402


def timed_loop(dev, chunks, sync=None):
    start = time.perf_counter()
    for counter in range(len(chunks)):
        chunks[counter] = dev.apply(chunks[counter])
    if sync:
        sync()
    stop = time.perf_counter()
    return {"total_ms": round((stop - start) * 1000, 2), "ms_per_chunk": round((stop - start) * 1000 / len(chunks), 5)}


def cpu_section(seconds):
    pyAudioDspTools.config.initialize(44100, 512)
    from pyAudioDspTools.Generators import CreateSinewave
    from pyAudioDspTools.Utility import MakeChunks, CombineChunks
    from pyAudioDspTools.EffectCompressor import CreateCompressor
    from pyAudioDspTools.EffectGate import CreateGate
    from pyAudioDspTools.EffectDelay import CreateDelay
    from pyAudioDspTools.EffectFFTFilter import CreateHighCutFilter, CreateLowCutFilter
    from pyAudioDspTools.EffectEQ3BandFFT import CreateEQ3BandFFT
    from pyAudioDspTools.EffectSoftClipper import CreateSoftClipper
    from pyAudioDspTools.EffectHardDistortion import CreateHardDistortion
    from pyAudioDspTools.EffectTremolo import CreateTremolo
    from pyAudioDspTools.EffectSaturator import CreateSaturator
    sine_chunked = MakeChunks(copy.deepcopy(CreateSinewave(1000, 44100 * seconds)))
    devices = [("Saturator", CreateSaturator()), ("Compressor", CreateCompressor()), ("Delay", CreateDelay()), ("Tremolo", CreateTremolo()),
               ("Hard Distortion", CreateHardDistortion()), ("Gate", CreateGate()), ("Lowcut FFT Filter", CreateLowCutFilter(200)),
               ("Highcut FFT Filter", CreateHighCutFilter(8000)), ("3 Band EQ FFT Version", CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5)),
               ("Soft Clipper", CreateSoftClipper())]
    for _, dev in devices:  # first call of a device: kernel set-up, not part of the loop's steady state
        dev.apply(sine_chunked[0].copy())
        if hasattr(dev, "reset"):
            dev.reset()
    out = {"chunks": len(sine_chunked), "chunk_size": 512}
    for name, dev in devices:
        out[name] = timed_loop(dev, sine_chunked)
    out["combined_peak"] = float(abs(CombineChunks(sine_chunked)).max())
    del devices, dev
    gc.collect()  # the ten devices' GPU memory is released HERE (a destroy waits for the device), not by a collection in the middle of the next section's timed loops
    return out


def gpu_section(seconds):
    import numpy
    import torch
    pyAudioDspTools.config.initialize(44100, 88200, use_gpu=True)
    from pyAudioDspTools.Generators import CreateSinewave
    from pyAudioDspTools.Utility import MakeChunks
    from pyAudioDspTools.EffectFFTFilterGPU import CreateHighCutFilterGPU, CreateLowCutFilterGPU
    from pyAudioDspTools.EffectEQ3BandFFTGPU import CreateEQ3BandFFTGPU
    sine_chunked = torch.from_numpy(numpy.array(MakeChunks(copy.deepcopy(CreateSinewave(1000, 44100 * seconds))))).cuda()
    devices = [("Lowcut FFT Filter", CreateLowCutFilterGPU(200)), ("Highcut FFT Filter", CreateHighCutFilterGPU(8000)),
               ("3 Band EQ FFT Version", CreateEQ3BandFFTGPU(100, 2, 700, -4, 8000, 5))]
    for _, dev in devices:
        dev.apply(sine_chunked[0].clone())
        dev.reset()
    out = {"chunks": len(sine_chunked), "chunk_size": 88200, "carrier": "torch CUDA tensor (cupy array in the reference)"}
    for name, dev in devices:
        out[name] = timed_loop(dev, sine_chunked, torch.cuda.synchronize)
    out["peak"] = float(sine_chunked.abs().max())
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=60, help="length of the sine (the reference's scripts: 60)")
    args = ap.parse_args()
    print(json.dumps({"ModuleTests.py": cpu_section(args.seconds), "ModuleTestsGPU.py": gpu_section(args.seconds)}, indent=1))
