"""Randomised differential test of the FFT engines against the exact (float64 direct-sum) engine on the GPU: random chunk
sizes (powers of two and arbitrary multiples of 4), filter kinds (cut filters, EQ composite, fused chain with and without
end-tap trimming, arbitrary asymmetric kernels), channel counts, transform lengths, kept-block sizes and call patterns
(single-step, multi-step, zero-copy ring, host buffers) - every output sample of every channel, 1e-5 of full scale.
Seeds are fixed: a failure names its case.  Run with -m gpu on MI355X."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("ADSP_FUZZ_SCALE", "1"))  # ADSP_FUZZ_SCALE=10: ten times as many seeded cases (soak runs)

POW2 = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
OTHER = [20, 100, 360, 1000, 1920, 3000, 4400, 12000]


def make_case(seed):
    from pyaudiodsptools_amd import FirStream, design
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice(POW2 if rng.random() < 0.7 else OTHER))
    fs = int(rng.choice([44100, 48000, 96000]))
    kind = str(rng.choice(["lowcut", "highcut", "eq", "chain", "chain_full", "random"]))
    lc = FirStream(design.lowcut_kernel(float(rng.uniform(50, 2000)), fs, n), n)
    hc = FirStream(design.highcut_kernel(float(rng.uniform(3000, 0.45 * fs)), fs, n), n)
    eq = FirStream(design.eq3_composite(100, float(rng.uniform(-6, 6)), 700, float(rng.uniform(-6, 6)), 8000, float(rng.uniform(-6, 6)), fs, n), n)
    if kind == "lowcut":
        fir = lc
    elif kind == "highcut":
        fir = hc
    elif kind == "eq":
        fir = eq
    elif kind in ("chain", "chain_full"):
        if n > 8192:  # the fused chain must fit one 32768-point transform
            n = 4096
            return make_case(seed + 5000)
        fir = lc.then(eq).then(hc)
        if kind == "chain":
            fir = fir.trimmed()
    else:  # an arbitrary (asymmetric) kernel with a random delay: exercises shift / lookback arithmetic
        m = int(rng.integers(3, max(4, n // 2)))
        taps = rng.normal(size=m) * np.hanning(m + 2)[1:-1]
        fir = FirStream(taps / np.abs(taps).sum(), n, latency_chunks=1, lookahead=int(rng.integers(0, max(1, n // 4))))
    channels = int(rng.choice([1, 2, 3, 5, 17, 64, 70]))
    steps = int(rng.integers(4, 10))
    return rng, n, fir, channels, steps, kind


@pytest.mark.parametrize("seed", range(200 * SCALE))
def test_random_geometry_and_call_pattern_against_the_exact_engine(seed):
    import torch
    import pyaudiodsptools_amd as adsp
    from pyaudiodsptools_amd import FirEngine, design
    rng, n, fir, channels, steps, kind = make_case(seed)
    if not design.fits_one_transform(fir):
        pytest.skip("kernel longer than one transform (covered by the partitioned-engine tests)")
    opt = str(rng.choice(["stream", "batch"]))
    mult = 0
    if n in POW2 and rng.random() < 0.3:
        try:
            design.overlap_save_geometry(fir, 4, opt)
            mult = 4
        except ValueError:
            mult = 0
    ring = int(rng.choice([0, 0, 7]))
    geo = design.overlap_save_geometry(fir, mult, opt)
    if ring and ring < geo.history_chunks + 1:
        ring = geo.history_chunks + 1
    eng = FirEngine(fir, channels=channels, fft_mult=mult, optimize_for=opt, ring_slots=ring)
    if n in POW2 and rng.random() < 0.5:  # a smaller kept block (multiple of N/4) for the multi-step launches
        g = n // 4
        eng.set_block_outputs(int(rng.integers(1, eng.geometry.max_block_outputs // g + 1)) * g)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(seed))
    y = torch.full_like(x, float("nan"))
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    s = torch.cuda.current_stream().cuda_stream
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    k = 0
    while k < steps:
        op = str(rng.choice(["dev1", "devk", "ring", "host"]))
        cnt = 1 if op in ("dev1", "ring") else int(rng.integers(1, steps - k + 1))
        if op == "host":
            torch.cuda.synchronize()
            y[k:k + cnt] = torch.from_numpy(eng.apply_host(x[k:k + cnt].cpu().numpy())).cuda()
        elif op == "ring":
            slot = eng.ring_acquire()
            assert hip.hipMemcpyAsync(slot, x[k].data_ptr(), channels * n * 4, 3, s) == 0
            eng.apply_ring(y[k], s)
        else:
            eng.apply_device(x[k:k + cnt], y[k:k + cnt], cnt, s)
        k += cnt
    torch.cuda.synchronize()
    ex = adsp.ExactFirEngine(fir, channels=channels)
    truth = torch.empty_like(x)
    ex.apply_device(x, truth, steps, s)
    torch.cuda.synchronize()
    scale = max(float(truth.abs().max()), 1e-3)
    err = float((y - truth).abs().max())
    what = f"seed {seed}: N={n} {kind} taps={len(fir.taps)} C={channels} steps={steps} F={eng.geometry.fft_size} V={eng.block_outputs} {opt} ring={ring}"
    assert bool(torch.isfinite(y).all()), what
    assert err <= 1e-5 * max(scale, 0.1), f"{what}: max|d| = {err:.3e}, scale {scale:.3e}"


@pytest.mark.parametrize("seed", range(60 * SCALE))
def test_random_int16_engines_against_the_exact_engine(seed):
    """int16 PCM batches (the S16 kernel instantiations of every plan, the large half-exchange ones included): never more
    than one LSB from the exact engine's int16 stream, and rarely."""
    import torch
    import pyaudiodsptools_amd as adsp
    from pyaudiodsptools_amd import FirEngine, design
    rng, n, fir, channels, steps, kind = make_case(300 + seed)
    if not design.fits_one_transform(fir):
        pytest.skip("kernel longer than one transform")
    opt = str(rng.choice(["stream", "batch"]))
    eng = FirEngine(fir, channels=channels, optimize_for=opt, sample_format="s16")
    amp = 8000 if kind in ("eq", "chain", "chain_full") else 30000  # EQ gains up to +6 dB per band: keep clear of int16 overflow
    x = torch.randint(-amp, amp, (steps, channels, n), device="cuda", dtype=torch.int16,
                      generator=torch.Generator(device="cuda").manual_seed(seed))
    y = torch.zeros_like(x)
    s = torch.cuda.current_stream().cuda_stream
    k = 0
    while k < steps:
        cnt = int(rng.integers(1, steps - k + 1))
        eng.apply_device(x[k:k + cnt], y[k:k + cnt], cnt, s)
        k += cnt
    ex = adsp.ExactFirEngine(fir, channels=channels, sample_format="s16")
    truth = torch.empty_like(x)
    ex.apply_device(x, truth, steps, s)
    torch.cuda.synchronize()
    d = (y.int() - truth.int()).abs()
    what = f"seed {seed}: N={n} {kind} taps={len(fir.taps)} C={channels} steps={steps} F={eng.geometry.fft_size} {opt}"
    assert int(d.max()) <= 1, f"{what}: {int(d.max())} LSB"
    assert float((d != 0).float().mean()) <= 0.01, f"{what}: {100 * float((d != 0).float().mean()):.2f} % of the samples differ"


@pytest.mark.parametrize("seed", range(60 * SCALE))
def test_random_fused_volume_and_accumulate_against_the_exact_engine(seed):
    """The EPI kernel instantiations of every plan: a fused VolumeChange with clipping on the output registers, and the
    accumulating output modes (add to what the buffer holds; add and clip = MixSignals), against the exact engine."""
    import torch
    import pyaudiodsptools_amd as adsp
    from pyaudiodsptools_amd import FirEngine, design
    rng, n, fir, channels, steps, kind = make_case(600 + seed)
    if not design.fits_one_transform(fir):
        pytest.skip("kernel longer than one transform")
    opt = str(rng.choice(["stream", "batch"]))
    eng = FirEngine(fir, channels=channels, optimize_for=opt)
    db = float(rng.uniform(-3, 9))
    mode = int(rng.choice([0, 1, 2]))
    adsp.config.initialize(44100, n)
    eng.set_epilogue(adsp.CreateVolumeChange(db))
    eng.set_accumulate(mode)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=gen)
    base = torch.empty_like(x).uniform_(-0.5, 0.5, generator=gen)
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    y = base.clone()
    s = torch.cuda.current_stream().cuda_stream
    k = 0
    while k < steps:
        cnt = int(rng.integers(1, steps - k + 1))
        eng.apply_device(x[k:k + cnt], y[k:k + cnt], cnt, s)
        k += cnt
    ex = adsp.ExactFirEngine(fir, channels=channels)
    truth = torch.empty_like(x)
    ex.apply_device(x, truth, steps, s)
    torch.cuda.synchronize()
    want = (truth * (10 ** (db / 20))).clamp(-1, 1)   # VolumeChange clips (Utility.py:189-194)
    if mode >= 1:
        want = want + base
    if mode == 2:
        want = want.clamp(-1, 1)
    what = f"seed {seed}: N={n} {kind} taps={len(fir.taps)} C={channels} steps={steps} F={eng.geometry.fft_size} {opt} mode={mode} {db:.2f} dB"
    scale = max(float(truth.abs().max()) * 10 ** (db / 20), 0.1)
    assert float((y - want).abs().max()) <= 1e-5 * max(scale, 1.0), f"{what}: {float((y - want).abs().max()):.3e}"


UNALIGNED = [5, 6, 10, 22, 30, 101, 250, 1001, 1002, 1999, 3001, 4410, 9999]


@pytest.mark.parametrize("seed", range(60 * SCALE))
def test_random_unaligned_chunk_sizes_against_the_exact_engine(seed):
    """Round 4: chunk sizes that are NOT multiples of 4 (or shorter than 16 samples) - the dword-access form of the generic kernel:
    cut filters with even and odd lengths, EQ composites, arbitrary kernels with random delays, ragged channels, every call
    pattern incl. the zero-copy ring and host buffers, with and without the accumulating output."""
    import torch
    import pyaudiodsptools_amd as adsp
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice(UNALIGNED))
    fs = int(rng.choice([44100, 48000, 96000]))
    kind = str(rng.choice(["lowcut", "highcut", "eq", "random"]))
    if kind == "lowcut":
        fir = FirStream(design.lowcut_kernel(float(rng.uniform(50, 2000)), fs, n), n)
    elif kind == "highcut":
        fir = FirStream(design.highcut_kernel(float(rng.uniform(3000, 0.45 * fs)), fs, n), n)
    elif kind == "eq":
        fir = FirStream(design.eq3_composite(100, float(rng.uniform(-6, 6)), 700, float(rng.uniform(-6, 6)), 8000, float(rng.uniform(-6, 6)), fs, n), n)
    else:
        m = int(rng.integers(1, max(2, n // 2)))
        taps = rng.normal(size=m)
        fir = FirStream(taps / np.abs(taps).sum(), n, latency_chunks=1, lookahead=int(rng.integers(0, max(1, n // 4))))
    if not design.fits_one_transform(fir):
        pytest.skip("kernel longer than one transform")
    channels = int(rng.choice([1, 2, 3, 5, 17, 64, 70]))
    steps = int(rng.integers(4, 12))
    opt = str(rng.choice(["stream", "batch"]))
    geo = design.overlap_save_geometry(fir, 0, opt)
    ring = int(rng.choice([0, 0, geo.history_chunks + 3]))
    eng = FirEngine(fir, channels=channels, optimize_for=opt, ring_slots=ring)
    acc = bool(rng.random() < 0.3)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=gen)
    base = torch.empty_like(x).uniform_(-0.5, 0.5, generator=gen) if acc else torch.full_like(x, float("nan"))
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    y = base.clone()
    if acc:
        eng.set_accumulate(1)
    s = torch.cuda.current_stream().cuda_stream
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    k = 0
    while k < steps:
        op = str(rng.choice(["dev1", "devk", "ring", "host"]))
        cnt = 1 if op in ("dev1", "ring") else int(rng.integers(1, steps - k + 1))
        if op == "host":
            torch.cuda.synchronize()
            buf = y[k:k + cnt].cpu().numpy().copy()   # (with accumulate on, the host call adds to what `out` holds)
            lib_out = np.ascontiguousarray(buf)
            from pyaudiodsptools_amd import _capi
            xin = np.ascontiguousarray(x[k:k + cnt].cpu().numpy())
            _capi.check(eng._lib.adsp_apply_host(eng._h, xin.ctypes.data_as(ctypes.c_void_p), lib_out.ctypes.data_as(ctypes.c_void_p), cnt))
            y[k:k + cnt] = torch.from_numpy(lib_out).cuda()
        elif op == "ring":
            slot = eng.ring_acquire()
            assert hip.hipMemcpyAsync(slot, x[k].data_ptr(), channels * n * 4, 3, s) == 0
            eng.apply_ring(y[k], s)
        else:
            eng.apply_device(x[k:k + cnt], y[k:k + cnt], cnt, s)
        k += cnt
    torch.cuda.synchronize()
    ex = adsp.ExactFirEngine(fir, channels=channels)
    truth = torch.empty_like(x)
    ex.apply_device(x, truth, steps, s)
    torch.cuda.synchronize()
    want = truth + base if acc else truth
    scale = max(float(truth.abs().max()), 1e-3)
    what = f"seed {seed}: N={n} {kind} taps={len(fir.taps)} C={channels} steps={steps} F={eng.geometry.fft_size} V={eng.block_outputs} {opt} ring={ring} acc={acc}"
    assert bool(torch.isfinite(y).all()), what
    assert float((y - want).abs().max()) <= 1e-5 * max(scale, 0.1), f"{what}: max|d| = {float((y - want).abs().max()):.3e}, scale {scale:.3e}"
