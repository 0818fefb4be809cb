"""GPU tests of the round-3 entry points: the RCCL broadcast behind the C ABI (adsp_bcast_spectrum), the stream
ordering of zero-copy ring steps issued on several streams, the sticky-hint fix of the set-spectrum calls and the
N > 1 control flow of bench.py with the real HIP engine.  Run with -m gpu on MI355X."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def adsp():
    import pyaudiodsptools_amd as pkg
    from pyaudiodsptools_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback by design)"
    return pkg


def _truth(adsp, fir, x):
    import torch
    ex = adsp.ExactFirEngine(fir, channels=x.shape[1])
    t = torch.empty_like(x)
    ex.apply_device(x, t, x.shape[0], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("kind", ["lowcut", "eq"])
def test_bcast_spectrum_world_of_one_keeps_filter_and_real_flag(adsp, kind):
    """adsp_bcast_spectrum with n = 1 (all a one-GPU box can run): librccl is opened, ncclCommInitAll builds the
    communicator, the broadcast runs on the engine's side stream and the engine rebuilds its tables from the device
    buffer - the real-spectrum flag, the kernel-reach hint and the outputs must survive."""
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    from pyaudiodsptools_amd.engine import broadcast_filter, rccl_version
    n, fs, channels, steps = 4096, 44100, 24, 5
    taps = design.lowcut_kernel(800, fs, n) if kind == "lowcut" else design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n)
    fir = FirStream(taps, n)
    eng = FirEngine(fir, channels=channels)
    was_real = eng.real_spectrum
    assert was_real == (kind == "lowcut")
    assert rccl_version() > 0
    broadcast_filter([eng], root=0)
    broadcast_filter([eng], root=0)  # the cached communicator
    assert eng.real_spectrum == was_real
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(5))
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for k in range(steps):  # single-step launches: the path that uses the kernel-reach hint
        eng.apply_device(x[k], y[k], 1, s)
    torch.cuda.synchronize()
    t = _truth(adsp, fir, x)
    assert float((y - t).abs().max()) <= 1e-5 * float(t.abs().max())


def test_bcast_spectrum_argument_errors(adsp):
    from pyaudiodsptools_amd import FirEngine, FirStream, _capi, design
    lib = _capi.load()
    n, fs = 512, 44100
    fir = FirStream(design.lowcut_kernel(300, fs, n), n)
    a, b = FirEngine(fir, channels=3), FirEngine(fir, channels=5)
    arr = (ctypes.c_void_p * 2)(a._h, b._h)
    assert lib.adsp_bcast_spectrum(arr, 2, 0) == _capi.ADSP_ERR_ARG and b"share device" in lib.adsp_last_error()
    assert lib.adsp_bcast_spectrum(arr, 1, 1) == _capi.ADSP_ERR_ARG
    assert lib.adsp_bcast_spectrum(arr, 0, 0) == _capi.ADSP_ERR_ARG
    assert lib.adsp_bcast_spectrum(None, 1, 0) == _capi.ADSP_ERR_ARG
    # a root without a spectrum: raw engine, never given one
    geo = a.geometry
    cfg = _capi.AdspConfig(0, n, 2, geo.fft_size, geo.history_chunks, geo.lookback, geo.out_offset, 0, _capi.ADSP_FORMAT_F32)
    h = ctypes.c_void_p(None)
    assert lib.adsp_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    one = (ctypes.c_void_p * 1)(h)
    assert lib.adsp_bcast_spectrum(one, 1, 0) == _capi.ADSP_ERR_STATE
    lib.adsp_destroy(h)


def test_set_spectrum_forgets_the_kernel_reach_hint(adsp):
    """the reach hint described the previous kernel.  A causal kernel (reach 0) followed, through the public
    upload_spectrum, by a zero-phase one (taps at negative circular indices) must not lose the window tail."""
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    n, fs, channels, steps = 1024, 44100, 7, 6
    sym = FirStream(design.lowcut_kernel(500, fs, n), n)
    eng = FirEngine(sym, channels=channels)
    geo = eng.geometry
    assert geo.shift < 0, "the cut filter is placed zero-phase: taps at negative circular indices"
    spec = eng.spectrum.copy()
    eng._lib.adsp_set_kernel_reach(eng._h, 0)  # what a previous causal kernel would have left behind
    eng.upload_spectrum(spec)                   # public path without a reach: must fall back to whole windows
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(3))
    y = torch.empty_like(x)
    for k in range(steps):
        eng.apply_device(x[k], y[k], 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = _truth(adsp, sym, x)
    assert float((y - t).abs().max()) <= 1e-5 * float(t.abs().max())


@pytest.mark.parametrize("n,kind,extra_slots,acquire_with_stream", [
    (512, "eq", 1, True), (512, "eq", 2, True), (4096, "lowcut", 1, True), (4096, "lowcut", 2, True),
    (512, "short", 1, True),      # history_chunks == 1
    (512, "eq", 2, False),        # plain adsp_ring_acquire: the host waits instead of the stream
    (4096, "lowcut", 3, True),
])
def test_two_stream_ring_steps_are_ordered_by_the_library_under_skew(adsp, n, kind, extra_slots, acquire_with_stream):
    """consecutive ring steps on two streams depend on each other through the ring - step k reads
    the slot producer k-1 filled on the OTHER stream, producer k overwrites a slot kernel k-3 (other stream) read.  One
    of the streams is held back by a spin kernel before some producers, so an unordered implementation filters stale or
    half-overwritten history; the library's per-step events must order it (history + 1 slots included)."""
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    fs, channels, steps = 44100, 64, 36
    if kind == "lowcut":
        fir = FirStream(design.lowcut_kernel(500, fs, n), n)
    elif kind == "eq":
        fir = FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n), n)
    else:
        fir = FirStream(np.random.default_rng(1).standard_normal(40) / 40, n, 1, n - 100)  # delay 100: the window reaches one chunk back
    from pyaudiodsptools_amd.design import overlap_save_geometry
    hist = overlap_save_geometry(fir, 0, "stream").history_chunks
    if kind == "short":
        assert hist == 1
    eng = FirEngine(fir, channels=channels, ring_slots=hist + extra_slots)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(n + extra_slots))
    y = torch.full_like(x, float("nan"))
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    for k in range(steps):
        st = streams[k % 2]
        if k % 5 in (1, 2):  # hold this stream back ~0.5 ms: the other one runs ahead unless it is ordered
            with torch.cuda.stream(st):
                torch.cuda._sleep(1_000_000)
        slot = eng.ring_acquire(st) if acquire_with_stream else eng.ring_acquire()
        assert hip.hipMemcpyAsync(slot, x[k].data_ptr(), channels * n * 4, 3, st.cuda_stream) == 0  # the producer
        eng.apply_ring(y[k], st)
    torch.cuda.synchronize()
    t = _truth(adsp, fir, x)
    assert bool(torch.isfinite(y).all())
    assert float((y - t).abs().max()) <= 1e-5 * float(t.abs().max())
    # back to one stream, then a multi-step launch on a third one: the step record starts over, nothing is lost
    y2 = torch.empty_like(x[:4])
    third = torch.cuda.Stream()
    eng.apply_device(x[:4], y2, 4, third.cuda_stream)
    third.synchronize()
    ex = adsp.ExactFirEngine(fir, channels=channels)
    xx = torch.cat([x, x[:4]])
    tt = torch.empty_like(xx)
    ex.apply_device(xx, tt, xx.shape[0], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert float((y2 - tt[steps:]).abs().max()) <= 1e-5 * float(tt.abs().max())


def test_set_fir_live_accepts_a_torch_stream(adsp):
    """engine.set_fir(live=True, stream=torch.cuda.Stream) used to raise TypeError in _ptr."""
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    n, fs = 512, 44100
    eng = FirEngine(FirStream(design.lowcut_kernel(300, fs, n), n), channels=4)
    st = torch.cuda.Stream()
    eng.set_fir(FirStream(design.lowcut_kernel(900, fs, n), n), stream=st, live=True)
    x = torch.ones((4, n), device="cuda")
    y = torch.empty_like(x)
    eng.apply_device(x, y, 1, st)
    st.synchronize()
    assert bool(torch.isfinite(y).all())


def test_bench_two_ranks_on_one_gpu_real_engine():
    """the N > 1 control flow of bench.py (process group, spectrum broadcast, barrier, max over ranks,
    one JSON line from rank 0, the extra 8-GPU configs, ranks_seen, the cross-rank spectrum checksum) with the REAL HIP
    engine: two gloo ranks share the one GPU of the box."""
    env = dict(os.environ, ADSP_BENCH_SINGLE_DEVICE="1", ADSP_BENCH_BACKEND="gloo", ADSP_BENCH_SMALL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--prewarm-ms", "20"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["spectrum_checksum"]["equal_on_all_ranks"] is True
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    cfgs = d["configs"]
    assert set(cfgs) >= {"config4_highcut_8192ch_x_4096", "config5_chain_4096ch_x_8192_96k"}
    for c in cfgs.values():
        assert c["value"] > 0 and c["n_gpus"] == 2 and c["roofline"]["frac"] > 0


def test_bench_single_process_mode_one_gpu():
    """bench.py --single-process: N engines in ONE process, filter shared by adsp_bcast_spectrum (no torch.distributed);
    with one GPU that is one engine, and the line says which carrier moved the spectrum."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--prewarm-ms", "20", "--no-cpu-baseline", "--no-latency", "--no-stream-extra", "--no-configs", "--chunks-per-step", "6", "--channels", "512"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"].startswith("channel-shard")
    assert d["spectrum_carrier"].startswith("adsp_bcast_spectrum")
    assert d["value"] > 0


@pytest.mark.parametrize("kind,fmt,channels", [
    ("lowcut", "f32", 5), ("highcut", "f32", 64), ("asym", "f32", 3), ("asym_long", "f32", 17), ("lowcut", "s16", 6),
    ("lowcut_epi", "f32", 9), ("lowcut_batchcalls", "f32", 4),
])
def test_three_times_power_of_two_plan_M3072(adsp, kind, fmt, channels):
    """F = 1.5 N = 6144 (M = 3072 = 3 * 2^10; radices 16 x 16 x 12, one wave per transform): what single-step launches of the
    cut filters at N = 4096 run on.  Real and complex spectrum stages, int16 samples, fused effect, ragged channel counts,
    single- and multi-step launches, against the float64 direct sum of the exact engine."""
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design, effects
    n, fs, steps = 4096, 44100, 7
    rng = np.random.default_rng(len(kind) + channels)
    if kind.startswith("lowcut"):
        fir = FirStream(design.lowcut_kernel(800, fs, n), n)
    elif kind == "highcut":
        fir = FirStream(design.highcut_kernel(8000, fs, n), n)
    elif kind == "asym":
        fir = FirStream(rng.standard_normal(300) / 100, n, 1, 700)
    else:
        taps = rng.standard_normal(2000) * np.hanning(2002)[1:-1]
        fir = FirStream(taps / np.abs(taps).sum(), n, 1, 1000)
    eng = FirEngine(fir, channels=channels, fft_mult=1.5, sample_format=fmt)
    assert eng.geometry.fft_size == 6144 and eng.plan["complex_points"] == 3072
    assert eng.real_spectrum == (kind in ("lowcut", "highcut", "lowcut_epi", "lowcut_batchcalls"))
    assert FirEngine(fir, channels=1, sample_format=fmt).geometry.fft_size == 8192, "opt-in: the default stays the 2N transform"
    eff = None
    if kind == "lowcut_epi":
        eff = effects.CreateSoftClipper()
        eng.set_epilogue(eff)
    gen = torch.Generator(device="cuda").manual_seed(channels)
    if fmt == "s16":
        x = torch.randint(-20000, 20000, (steps, channels, n), device="cuda", dtype=torch.int16, generator=gen)
    else:
        x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=gen)
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    if kind == "lowcut_batchcalls":
        eng.apply_device(x[:3], y[:3], 3, s)  # multi-step launches of the 1.5 N engine keep N per transform
        eng.apply_device(x[3:], y[3:], steps - 3, s)
    else:
        for k in range(steps):
            eng.apply_device(x[k], y[k], 1, s)
    torch.cuda.synchronize()
    ex = adsp.ExactFirEngine(fir, channels=channels, sample_format=fmt)
    t = torch.empty_like(x)
    ex.apply_device(x, t, steps, s)
    torch.cuda.synchronize()
    if fmt == "s16":
        d = (y.int() - t.int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
        return
    if eff is not None:
        t = torch.from_numpy(eff.apply(t.cpu().numpy().reshape(-1)).reshape(t.shape)).cuda()
    assert bool(torch.isfinite(y).all())
    assert float((y - t).abs().max()) <= 1e-5 * max(float(t.abs().max()), 0.1)


def _copy_fn():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    return hip.hipMemcpyAsync


@pytest.mark.parametrize("n,kind", [(512, "eq"), (4096, "lowcut"), (1024, "eq")])
def test_resident_ring_launch_waits_for_a_producer_that_starts_later(adsp, n, kind):
    """adsp_apply_ring_resident: ONE launch covers many ring steps; its workgroups wait on the sequence word
    the producer stream bumps.  The consumer is launched BEFORE any input exists, the producer (device copies on a second
    stream) starts 30 ms later; a second and third launch wrap the ring (the producer must wait for the launches that
    still read the slots it overwrites) with the producer partly ahead of, partly behind the consumer launch."""
    import time
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    fs, channels = 44100, 40
    taps = design.lowcut_kernel(500, fs, n) if kind == "lowcut" else design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n)
    fir = FirStream(taps, n)
    hist = design.overlap_save_geometry(fir, 0, "stream").history_chunks
    per, launches = 10, 3
    eng = FirEngine(fir, channels=channels, ring_slots=per + hist)
    steps = per * launches
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(n))
    y = torch.full_like(x, float("nan"))
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    cons, prod = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    copy = _copy_fn()

    def produce(k):
        slot = eng.ring_produce_begin(prod)
        assert copy(slot, x[k].data_ptr(), channels * n * 4, 3, prod.cuda_stream) == 0
        eng.ring_produce_end(prod)
    # launch 1: consumer first, producer 30 ms later
    eng.apply_ring_resident(y[:per], per, cons)
    time.sleep(0.03)
    assert not cons.query(), "the resident launch must still be waiting for its input"
    for k in range(per):
        produce(k)
    # launch 2: producer ahead by 4 steps, consumer launched, rest produced afterwards
    for k in range(per, per + 4):
        produce(k)
    eng.apply_ring_resident(y[per:2 * per], per, cons)
    for k in range(per + 4, 2 * per):
        produce(k)
    # launch 3: everything published before the launch
    for k in range(2 * per, 3 * per):
        produce(k)
    eng.apply_ring_resident(y[2 * per:], per, cons)
    torch.cuda.synchronize()
    assert not eng.ring_resident_timed_out()
    t = _truth(adsp, fir, x)
    assert bool(torch.isfinite(y).all())
    assert float((y - t).abs().max()) <= 1e-5 * float(t.abs().max())
    # the per-step calls are refused until the ring has left resident mode
    with pytest.raises(adsp._capi.AdspError):
        eng.apply_ring(y[0], cons)
    eng.ring_reset_order()
    z = torch.empty_like(x[:2])
    for k in range(2):  # the stream continues where the resident launches left it
        slot = eng.ring_acquire()
        assert copy(slot, x[k].data_ptr(), channels * n * 4, 3, cons.cuda_stream) == 0
        eng.apply_ring(z[k], cons)
    torch.cuda.synchronize()
    ex = adsp.ExactFirEngine(fir, channels=channels)
    xx = torch.cat([x, x[:2]])
    tt = torch.empty_like(xx)
    ex.apply_device(xx, tt, xx.shape[0], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert float((z - tt[steps:]).abs().max()) <= 1e-5 * float(tt.abs().max())


def test_resident_ring_launch_without_a_producer_times_out_instead_of_hanging(adsp):
    import time
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    n, fs, channels = 512, 44100, 8
    fir = FirStream(design.lowcut_kernel(300, fs, n), n)
    eng = FirEngine(fir, channels=channels, ring_slots=8)
    eng.ring_resident_timeout(40.0)
    y = torch.full((4, channels, n), float("nan"), device="cuda")
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    with pytest.raises(adsp._capi.AdspError):
        eng.apply_ring_resident(y, 7, None)  # more steps than ring_slots - history
    t0 = time.perf_counter()
    eng.apply_ring_resident(y, 4, None)
    torch.cuda.synchronize()
    assert 0.03 < time.perf_counter() - t0 < 5.0
    assert eng.ring_resident_timed_out() and not eng.ring_resident_timed_out()  # reported once
    assert bool(torch.isnan(y).all()), "a block that gave up writes nothing"
    eng.reset()
    x = torch.ones((channels, n), device="cuda")
    out = torch.empty_like(x)
    eng.apply_device(x, out, 1, None)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())


def test_example1_full_run_through_the_dropin_surface(adsp):
    """VERDICT r2 #4 / SURVEY 3a: Example1.py in full - MonoWavToNumpyFloat's conversion, MakeChunks (65 chunks, 1640
    padded zeros), CreateLowCutFilter(800).apply per chunk, CombineChunks - against the reference's merged output
    (every 8th sample, first and last chunk in full).  The ragged end: the last input chunk is never flushed."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "kat_example1_full.npz"))
    adsp.config.initialize(44100, 4096)
    full = g["pcm16"].astype(np.float32) / 32768  # Utility.py:236-237
    chunks = adsp.MakeChunks(full)
    assert len(chunks) == 65
    dev = adsp.CreateLowCutFilter(800)
    outs = [dev.apply(c) for c in chunks]
    merged = adsp.CombineChunks(outs)
    assert len(merged) == 266240 and merged.dtype == np.float32
    scale = float(np.abs(g["out_dec8"]).max())
    assert np.abs(merged[::8] - g["out_dec8"]).max() <= 1e-5 * scale
    assert np.abs(outs[0] - g["out_first_chunk"]).max() <= 1e-5 * scale
    assert np.abs(outs[-1] - g["out_last_chunk"]).max() <= 1e-5 * scale
    # the same file as ONE batched launch over all 65 chunks (what WavBank does for many files): identical stream
    eng = adsp.FirEngine(dev.fir, channels=1, optimize_for="batch")
    y = eng.apply_host(np.stack(chunks)[:, None, :])[:, 0, :].reshape(-1)
    assert np.abs(y[::8] - g["out_dec8"]).max() <= 1e-5 * scale


def _int16_mismatch(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), int((d > 0).sum()), d.size


def test_exact_fft_engine_on_the_reference_wav_fixtures(adsp):
    """a FAST exact int16 path.  sample_format "s16_f64" runs the same kernels in float64 with the reference's
    conversions to the letter: on the Example1 / Example2 WAV slices it must equal the float64 direct sum (ExactFirEngine)
    bit for bit, and differ from the reference's own int16 export no more than the direct sum does (one boundary sample)."""
    from pyaudiodsptools_amd import ExactFirEngine, FirEngine, FirStream, design
    n, fs = 4096, 44100
    fir = FirStream(design.lowcut_kernel(800, fs, n), n)
    g1 = np.load(os.path.join(ROOT, "tests", "golden", "kat_example1.npz"))
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "kat_example2.npz"))
    pcm1 = g1["pcm16_first8"].reshape(8, 1, n)
    pcm2 = np.ascontiguousarray(g2["pcm16_first4_stereo"].T.reshape(2, 4, n).transpose(1, 0, 2))
    for pcm, refs in ((pcm1, [g1["out_first8"]]), (pcm2, [g2["out_left"], g2["out_right"]])):
        for opt in ("batch", "stream"):
            eng = FirEngine(fir, channels=pcm.shape[1], sample_format="s16_f64", optimize_for=opt)
            y = eng.apply_host(pcm) if opt == "batch" else np.stack([eng.apply_host(pcm[k]) for k in range(pcm.shape[0])])
            ex = ExactFirEngine(fir, channels=pcm.shape[1], sample_format="s16")
            t = ex.apply_host(pcm)
            assert np.array_equal(y, t), _int16_mismatch(y, t)
            for c, ref in enumerate(refs):
                ref16 = (ref * 32767).astype(np.int16)  # Utility.py:306
                worst, count, size = _int16_mismatch(y[:, c].reshape(-1), ref16)
                assert worst <= 1 and count <= 1, (worst, count, size)


@pytest.mark.parametrize("seed", range(40))
def test_exact_fft_engine_random_cases_against_the_direct_sum(adsp, seed):
    """Random chunk sizes (powers of two and arbitrary multiples of 4: the generic-geometry kernel), filters, channel counts and
    call patterns in float64: never more than one LSB from the direct sum, and only where the float64 result straddles a
    float32 rounding boundary (expected about 1e-7 of the samples; the test allows 1e-5)."""
    import torch
    from pyaudiodsptools_amd import ExactFirEngine, FirEngine, FirStream, design
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 8192, 1000, 1920, 3000, 20 * 4, 12000]))
    fs = int(rng.choice([44100, 48000, 96000]))
    kind = str(rng.choice(["lowcut", "highcut", "eq", "chain", "asym"]))
    lc = FirStream(design.lowcut_kernel(float(rng.uniform(50, 2000)), fs, n), n)
    hc = FirStream(design.highcut_kernel(float(rng.uniform(3000, 0.45 * fs)), fs, n), n)
    eq = FirStream(design.eq3_composite(100, float(rng.uniform(-6, 6)), 700, float(rng.uniform(-6, 6)), 8000, float(rng.uniform(-6, 6)), fs, n), n)
    if kind == "chain" and n <= 8192:
        fir = lc.then(eq).then(hc).trimmed()
    elif kind == "asym":
        m = int(rng.integers(3, max(4, n // 2)))
        taps = rng.normal(size=m) * np.hanning(m + 2)[1:-1]
        fir = FirStream(taps / np.abs(taps).sum(), n, 1, int(rng.integers(0, max(1, n // 4))))
    else:
        fir = {"lowcut": lc, "highcut": hc, "eq": eq, "chain": eq}[kind]
    if not design.fits_one_transform(fir):
        pytest.skip("kernel longer than one transform")
    channels, steps = int(rng.choice([1, 2, 5, 17, 64])), int(rng.integers(3, 8))
    opt = str(rng.choice(["stream", "batch"]))
    eng = FirEngine(fir, channels=channels, sample_format="s16_f64", optimize_for=opt)
    x = torch.randint(-32768, 32768, (steps, channels, n), device="cuda", dtype=torch.int16, generator=torch.Generator(device="cuda").manual_seed(seed))
    y = torch.zeros_like(x)
    s = torch.cuda.current_stream().cuda_stream
    if rng.random() < 0.5:
        eng.apply_device(x, y, steps, s)
    else:
        for k in range(steps):
            eng.apply_device(x[k], y[k], 1, s)
    ex = ExactFirEngine(fir, channels=channels, sample_format="s16")
    t = torch.zeros_like(x)
    ex.apply_device(x, t, steps, s)
    torch.cuda.synchronize()
    worst, count, size = _int16_mismatch(y.cpu().numpy(), t.cpu().numpy())
    assert worst <= 1 and count <= max(1, int(1e-5 * size)), f"N={n} {kind} F={eng.geometry.fft_size} {opt}: {count} of {size} differ, worst {worst}"


def test_local_bank_one_process_real_engines(adsp):
    """dist.LocalFirBank on the GPUs of this process (one here): channel shards, adsp_bcast_spectrum, concurrent apply."""
    from pyaudiodsptools_amd import FirStream, design
    from pyaudiodsptools_amd.dist import LocalFirBank
    n, fs, channels, steps = 1024, 44100, 11, 5
    fir = FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n), n)
    bank = LocalFirBank(fir, channels, devices=[0])
    x = np.random.default_rng(2).uniform(-1, 1, (steps, channels, n)).astype(np.float32)
    y = bank.apply_host(x)
    ex = adsp.ExactFirEngine(fir, channels=channels)
    t = ex.apply_host(x)
    assert np.abs(y - t).max() <= 1e-5 * np.abs(t).max()
    bank.close()
