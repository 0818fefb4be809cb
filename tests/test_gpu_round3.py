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
        assert c["value"] > 0 and c["n_gpus"] == 2 and c["roofline_frac"] > 0


def test_bench_single_process_mode_one_gpu():
    """bench.py --single-process: N engines in ONE process, filter shared by adsp_bcast_spectrum (no torch.distributed);
    with one GPU that is one engine, and the line says which carrier moved the spectrum."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--prewarm-ms", "20", "--no-cpu-baseline", "--no-latency", "--no-stream-extra", "--chunks-per-step", "6", "--channels", "512"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"].startswith("channel-shard")
    assert d["spectrum_carrier"].startswith("adsp_bcast_spectrum")
    assert d["value"] > 0
