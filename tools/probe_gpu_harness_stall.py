#!/usr/bin/env python3
"""Probe behind examples/harness_timing.py's GPU section (round 6): per-call times of the three loops at chunk 88200 on device arrays, device
construction times, and calls slower than 0.3 ms.  Finding: no call of the loops is slow by itself; the 75 ms that one of the three loops
showed in some runs of the example was the garbage collector destroying the TEN devices of the example's first section (every destroy waits
for the device) in the middle of a timed loop - the example now collects between its sections."""
import sys, os, time, copy, numpy, torch
sys.path.insert(0, os.getcwd())
import pyaudiodsptools_amd.compat; pyaudiodsptools_amd.compat.install()
import pyAudioDspTools
pyAudioDspTools.config.initialize(44100, 88200, use_gpu=True)
from pyAudioDspTools.Generators import CreateSinewave
from pyAudioDspTools.Utility import MakeChunks
from pyAudioDspTools.EffectFFTFilterGPU import CreateHighCutFilterGPU, CreateLowCutFilterGPU
from pyAudioDspTools.EffectEQ3BandFFTGPU import CreateEQ3BandFFTGPU
for rnd in range(4):
    arr = torch.from_numpy(numpy.array(MakeChunks(copy.deepcopy(CreateSinewave(1000, 44100 * 60))))).cuda()
    t_c = time.perf_counter()
    devs = []
    for name, mk in (("LC", lambda: CreateLowCutFilterGPU(200)), ("HC", lambda: CreateHighCutFilterGPU(8000)), ("EQ", lambda: CreateEQ3BandFFTGPU(100, 2, 700, -4, 8000, 5))):
        t_c = time.perf_counter()
        devs.append((name, mk()))
        print("round", rnd, "create", name, "ms", round((time.perf_counter() - t_c) * 1000, 1))
    for name, dev in devs:
        t0 = time.perf_counter(); dev.apply(arr[0].clone()); torch.cuda.synchronize(); t1 = time.perf_counter(); dev.reset(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("  warm", name, "apply ms", round((t1 - t0) * 1000, 2), "reset ms", round((t2 - t1) * 1000, 2))
    for name, dev in devs:
        ts = []
        for i in range(len(arr)):
            t0 = time.perf_counter(); arr[i] = dev.apply(arr[i]); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); torch.cuda.synchronize(); tsync = time.perf_counter() - t0
        slow = [(i, round(t * 1e6)) for i, t in enumerate(ts) if t > 300e-6]
        print("  loop", name, "median us", round(sorted(ts)[len(ts) // 2] * 1e6, 1), "slow calls (index, us):", slow, "final sync us", round(tsync * 1e6))
