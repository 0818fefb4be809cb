import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pyaudiodsptools_amd as adsp
from pyaudiodsptools_amd import design
n, fs = 88200, 44100
for name, taps in (("lowcut", design.lowcut_kernel(800, fs, n)), ("eq3", design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n))):
    for C in (1, 2, 8):
        fir = adsp.FirStream(taps, n)
        eng = adsp.make_engine(fir, channels=C)
        x = torch.rand((C, n), device="cuda"); y = torch.empty_like(x)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5): eng.apply_device(x, y, 1, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): eng.apply_device(x, y, 1, s)
        e1.record(); torch.cuda.synchronize()
        dev_us = e0.elapsed_time(e1) * 1e3 / 20
        xh = x.cpu().numpy()
        for _ in range(5): eng.apply_host(xh)
        t0 = time.perf_counter()
        for _ in range(20): eng.apply_host(xh)
        host_us = (time.perf_counter() - t0) / 20 * 1e6
        print(name, "C", C, type(eng).__name__, "block", eng.block, "P", eng.partition.n_partitions, "device us/call %.1f" % dev_us, "host us/call %.1f" % host_us, flush=True)
        eng.close()
adsp.config.initialize(fs, n)
for mk in (lambda: adsp.CreateLowCutFilter(800), lambda: adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5)):
    dev = mk()
    xx = np.random.default_rng(0).uniform(-1, 1, n).astype(np.float32)
    for _ in range(5): dev.apply(xx)
    t0 = time.perf_counter()
    for _ in range(20): dev.apply(xx)
    print(type(dev).__name__, "apply us/call %.1f" % ((time.perf_counter() - t0) / 20 * 1e6), "fir delay", dev.fir.delay, "taps", len(dev.fir.taps), type(dev.engine).__name__)
    t0 = time.perf_counter()
    for _ in range(20): dev.engine.apply_host(xx.reshape(1, n))
    print("   engine.apply_host us/call %.1f" % ((time.perf_counter() - t0) / 20 * 1e6))
