"""Round 5: the uniformly partitioned engine for kernels longer than one transform (Example4's chunk 88200), the
counter-based input generator and its numpy twin, the bench line's new blocks (configs 4 / 5 at N = 1, the host oracle check, the loud
exit on a wrong world), the live-session guards.  Run with -m gpu on MI355X."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_parity, seeded_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def adsp():
    import pyaudiodsptools_amd as pkg
    from pyaudiodsptools_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback by design)"
    return pkg


def orc():
    from oracle import fftfilter_oracle as o
    return o


def _exact(adsp, fir, x, fmt="f32"):
    import torch
    ex = adsp.ExactFirEngine(fir, channels=x.shape[1], sample_format=fmt)
    t = torch.empty_like(x)
    ex.apply_device(x, t, x.shape[0], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ex.close()
    return t


# ---------------------------------------------------------------------------------------------------------------------
# 1. Uniformly partitioned engines (csrc/adsp_upols.hip)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("block", [8192, 16384])
@pytest.mark.parametrize("n,taps_len,latency,lookahead,channels,calls,max_steps", [
    (88200, 44099, 1, 22049, 3, [1, 1, 1], 1),        # Example4's low cut geometry (Example4.py:5, EffectFFTFilter.py:91-151), random taps
    (20000, 40000, 3, 20000, 9, [2, 1, 3], 2),        # kernel longer than two chunks, calls of several chunks, split into sub-calls
    (12000, 33000, 4, 16000, 67, [1, 2, 1, 1], 4),    # ragged channel count (the last XCD group is partly empty), odd delay (shift 0..3)
    (50000, 70001, 2, 12345, 1, [1, 1], 1),           # ONE channel, delay = 87655 -> 3 taps of kernel delay
    (50000, 300001, 1, 20000, 2, [1, 2, 1], 1),       # a kernel of six chunks: 37 / 19 partitions
])
def test_upols_engine_matches_the_float64_direct_sum(adsp, n, taps_len, latency, lookahead, channels, calls, max_steps, block):
    """Every output sample of every channel of a long-kernel stream against the float64 direct sum computed on the GPU
    (adsp_exact_*), and two channels against the oracle's direct_stream_convolution on the host; the same stream through
    PartitionedFirEngine (the round 1 - 4 form) agrees."""
    import torch
    rng = np.random.default_rng(n + taps_len)
    taps = rng.standard_normal(taps_len) * np.hanning(taps_len) / np.sqrt(taps_len) * 3.0
    fir = adsp.FirStream(taps, n, latency_chunks=latency, lookahead=lookahead)
    eng = adsp.UpolsFirEngine(fir, channels=channels, max_steps=max_steps, block=block)
    assert eng.block == block and adsp.UpolsFirEngine.block_sizes() == [8192, 16384]
    assert eng.partition.n_partitions == -(-(taps_len + eng.partition.shift) // block) and eng.partition.delay % 4 == 0
    steps = sum(calls)
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=g)
    y = torch.full_like(x, 7.0)
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    s = torch.cuda.current_stream().cuda_stream
    pos = 0
    for k in calls:
        eng.apply_device(x[pos:pos + k], y[pos:pos + k], k, s)
        pos += k
    torch.cuda.synchronize()
    t = _exact(adsp, fir, x)
    scale = float(t.abs().max())
    assert scale > 0.1 and float((y - t).abs().max()) <= 1e-5 * scale, float((y - t).abs().max()) / scale
    for c in sorted({0, channels - 1}) if taps_len <= 100000 else ():   # (the host's direct sum of the longest kernel would take minutes)
        ref = orc().direct_stream_convolution(taps, x[:, c].reshape(-1).cpu().numpy(), n, latency, lookahead)
        assert_parity(y[:, c].reshape(-1).cpu().numpy(), ref, what=f"oracle channel {c}")
    # the host path on a fresh stream, and reset
    eng.reset()
    yh = eng.apply_host(x[:calls[0]].cpu().numpy())
    assert np.abs(yh - y[:calls[0]].cpu().numpy()).max() <= 2e-6 * scale
    eng.close()
    if channels <= 9 and taps_len <= 100000:
        pe = adsp.PartitionedFirEngine(fir, channels=channels)
        yp = torch.empty_like(x)
        pe.apply_device(x, yp, steps, s)
        torch.cuda.synchronize()
        assert float((yp - t).abs().max()) <= 1e-5 * scale
        pe.close()


@pytest.mark.parametrize("block", [8192, 16384])
def test_upols_engine_int16_and_fused_effect(adsp, block):
    """int16 PCM batches (the WAV front end, Utility.py:233-238 / :295-312, fused like the ADSP_FORMAT_S16 engines: <= 1 LSB against the
    exact engine's int16 stream) and a stateless effect on the output registers (EffectSaturator.py:27-49 after the long filter)."""
    import torch
    from oracle import effects_oracle as fx
    n, taps_len = 30000, 36001
    rng = np.random.default_rng(5)
    taps = rng.standard_normal(taps_len) * np.hanning(taps_len) / np.sqrt(taps_len)
    fir = adsp.FirStream(taps, n, latency_chunks=2, lookahead=18000)
    pcm = torch.randint(-12000, 12000, (3, 5, n), device="cuda", dtype=torch.int16, generator=torch.Generator(device="cuda").manual_seed(9))
    eng = adsp.UpolsFirEngine(fir, channels=5, sample_format="s16", block=block)
    assert eng.block == block
    out = torch.empty_like(pcm)
    eng.apply_device(pcm, out, 3, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = _exact(adsp, fir, pcm, "s16")
    diff = (out.int() - want.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) <= 0.01
    eng.close()
    # fused saturator against the oracle's saturator applied to the exact stream
    x = torch.empty((2, 4, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(10))
    eng = adsp.UpolsFirEngine(adsp.FirStream(taps * 4.0, n, latency_chunks=2, lookahead=18000), channels=4, block=block)
    eng.set_epilogue(adsp.CreateSaturator())
    y = torch.empty_like(x)
    eng.apply_device(x, y, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = _exact(adsp, eng.fir, x).cpu().numpy()
    assert_parity(y.cpu().numpy(), fx.saturator(t), what="fused saturator")
    # (round 6: the tremolo fuses too - tests/test_gpu_round6.py::test_tremolo_behind_a_long_kernel; an int16 engine still takes no effect)
    eng.close()
    eng = adsp.UpolsFirEngine(fir, channels=2, sample_format="s16", block=block)
    with pytest.raises(adsp.AdspError):
        eng.set_epilogue(adsp.CreateSaturator())
    eng.close()


def test_upols_raw_abi_refusals(adsp):
    import ctypes
    from pyaudiodsptools_amd import _capi
    lib = _capi.load()
    b = lib.adsp_upols_block_size()
    spec = np.zeros((2, b + 1, 2), np.float32)
    h = ctypes.c_void_p(None)
    ok = dict(device_id=0, chunk_size=40000, n_channels=2, block_size=b, n_partitions=2, delay=20000, sample_format=0, max_steps=1)
    for bad in (dict(block_size=4096), dict(chunk_size=40002), dict(delay=b - 4), dict(delay=20002), dict(n_partitions=0), dict(sample_format=2),
                dict(max_steps=0), dict(n_channels=0), dict(device_id=99)):
        cfg = _capi.AdspUpolsConfig(**{**ok, **bad})
        assert lib.adsp_upols_create(ctypes.byref(cfg), spec.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) != 0, bad
        assert lib.adsp_last_error()
    cfg = _capi.AdspUpolsConfig(**ok)
    assert lib.adsp_upols_create(ctypes.byref(cfg), spec.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == 0
    assert lib.adsp_upols_apply_device(h, None, None, 1, None) != 0 and lib.adsp_upols_set_epilogue(h, 5, 0.4, 1e-4, 0.0) != 0  # tremolo: table length 0
    assert lib.adsp_upols_set_epilogue(h, 7, 0.0, 0.0, 0.0) != 0 and lib.adsp_upols_set_epilogue(h, 5, 0.4, 1e-4, 100.0) == 0 and lib.adsp_upols_set_epilogue(h, 0, 0.0, 0.0, 0.0) == 0
    import torch
    buf = torch.zeros((3, 2, 40000), device="cuda")   # in place, or overlapping by one chunk: refused
    assert lib.adsp_upols_apply_device(h, ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(buf.data_ptr()), 2, None) != 0 and b"overlap" in lib.adsp_last_error()
    assert lib.adsp_upols_apply_device(h, ctypes.c_void_p(buf[0].data_ptr()), ctypes.c_void_p(buf[1].data_ptr()), 2, None) != 0
    assert lib.adsp_upols_apply_device(h, ctypes.c_void_p(buf[0].data_ptr()), ctypes.c_void_p(buf[2].data_ptr()), 1, None) == 0
    torch.cuda.synchronize()
    lib.adsp_upols_destroy(h)


def test_large_host_batches_move_through_pipelined_pinned_staging(adsp):
    """adsp_apply_host on a real batch (EffectFFTFilter.py:49-75 for many channels and chunks per call) - slabs through
    double-buffered pinned staging, H2D / kernel / D2H overlapped.  The result equals the device-resident path's (every sample against
    the float64 direct sum), slab boundaries carry the history, a second call continues the stream, int16 batches and the
    one-piece fallback (ADSP_HOST_UNPIPELINED) agree."""
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    n, C, steps = 4096, 96, 37   # 58 MB per direction: 7 slabs of whole tiles, a ragged last one
    fir = FirStream(design.lowcut_kernel(800, 44100, n), n)
    x = torch.empty((steps, C, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(21))
    t = _exact(adsp, fir, x).cpu().numpy()
    xh = x.cpu().numpy()
    scale = float(np.abs(t).max())
    for optimize_for in ("batch", "stream"):
        eng = FirEngine(fir, channels=C, optimize_for=optimize_for)
        y = eng.apply_host(xh[:30])
        out = np.empty_like(xh[30:])
        y2 = eng.apply_host(xh[30:], out=out)   # a second call continues the stream; `out` is filled in place
        assert y2 is out and np.abs(np.concatenate([y, y2]) - t).max() <= 1e-5 * scale, optimize_for
        os.environ["ADSP_HOST_UNPIPELINED"] = "1"
        try:
            eng.reset()
            y3 = eng.apply_host(xh)
        finally:
            del os.environ["ADSP_HOST_UNPIPELINED"]
        assert np.abs(y3 - t).max() <= 1e-5 * scale
        with pytest.raises(ValueError):
            eng.apply_host(xh, out=np.empty((steps, C, n), np.float64))
        eng.close()
    pcm = torch.randint(-12000, 12000, (64, 200, n), device="cuda", dtype=torch.int16, generator=torch.Generator(device="cuda").manual_seed(22))
    want = _exact(adsp, fir, pcm, "s16").cpu().numpy().astype(np.int32)
    eng = FirEngine(fir, channels=200, sample_format="s16", optimize_for="batch")
    got = eng.apply_host(pcm.cpu().numpy()).astype(np.int32)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() <= 0.01
    eng.close()


# ---------------------------------------------------------------------------------------------------------------------
# 2. The counter-based generator and its numpy twin
# ---------------------------------------------------------------------------------------------------------------------
def test_synth_device_equals_its_numpy_twin_bit_for_bit(adsp):
    import torch
    from pyaudiodsptools_amd import synth
    for fmt, dt in (("f32", torch.float32), ("s16", torch.int16)):
        for (c0, t0, C, N, steps, amp) in ((0, 0, 5, 64, 3, 1.0), (4090, (1 << 33) + 12, 7, 4096, 2, 1.0), (17, 123456, 3, 1000, 4, 0.25)):
            d = torch.empty((steps, C, N), device="cuda", dtype=dt)
            synth.fill_device(d, 1234, c0, t0, C, N, steps, fmt, amp)
            torch.cuda.synchronize()
            want = synth.batch_host(1234, c0, C, t0, N, steps, fmt, amp)
            assert np.array_equal(d.cpu().numpy(), want), (fmt, c0, t0)
    u = synth.uniform_host(1234, 3, 0, 1 << 16)
    assert u.dtype == np.float32 and -1.0 <= u.min() < -0.99 and 0.99 < u.max() < 1.0 and abs(float(u.mean())) < 0.02
    assert abs(float(np.corrcoef(u[:-1], u[1:])[0, 1])) < 0.02 and abs(float(np.corrcoef(u, synth.uniform_host(1234, 4, 0, 1 << 16))[0, 1])) < 0.02


# ---------------------------------------------------------------------------------------------------------------------
# 3. bench.py: configs 4 / 5 in the N = 1 line, the oracle check, the loud exit
# ---------------------------------------------------------------------------------------------------------------------
def _bench(*extra, env=None, check=True):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--channels", "256", "--chunks-per-step", "12", "--prewarm-ms", "20", "--cpu-seconds", "0.2", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env={**os.environ, **(env or {})})
    if check:
        assert out.returncode == 0, out.stderr[-2000:]
    return out


def test_bench_n1_line_carries_configs_4_and_5_with_their_rooflines_and_the_oracle_check():
    out = _bench("--steps", "3", "--warmup", "1", "--no-stream-extra", "--no-latency", "--no-cpu-baseline")
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.strip()][-1])
    assert d["oracle_checked"] is True and 0.0 <= d["oracle_check"]["max_rel_err"] <= 1e-5 and len(d["oracle_check"]["channels"]) == 2
    assert d["oracle_check"]["samples"] >= 2 * 2 * 3 * 4096 and "direct_stream_convolution" in d["oracle_check"]["against"]
    for key, taps in (("config4_highcut_8192ch_x_4096", 2047), ("config5_chain_4096ch_x_8192_96k", 9401)):
        c = d["configs"][key]
        assert "error" not in c, c
        assert c["n_gpus"] == 1 and c["value"] > 0 and c["runs"] == 3 and len(c["runs_msamples_s"]) == 3 and f"{taps} taps" in c["workload"]
        assert c["parity_checked"] is True and c["parity_max_rel_err"] <= 1e-5 and c["oracle_max_rel_err"] <= 1e-5
        r = c["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["launches"] == c["steps"]
        assert abs(r["achieved"] * 1e9 - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6)) <= 2e-3 * r["achieved"] * 1e9
    assert d["data"].startswith("synthetic uniform(-1,1) float32")


def test_bench_long_kernel_figures(adsp):
    """The bench line's latency.long_kernels block (Example4's chunk 88200 through make_engine): on the uniformly partitioned engine,
    within 1e-5 of the float64 direct sum."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench
    f = bench.long_kernel_figures(torch.device("cuda", 0), channels=6, calls=3)
    for key, parts in (("lowcut_44099_taps", 6), ("eq3_88197_taps", 11)):
        r = f[key]
        assert r["engine"] == "UpolsFirEngine" and r["block"] == 8192 and r["partitions"] == parts
        assert r["us_per_call"] > 0 and r["msamples_s"] > 0 and 0 <= r["max_rel_err_vs_float64_direct_sum"] <= 1e-5
    assert "88200" in f["workload"] and all(v > 0 for v in f["numpy_api_1ch_us_per_call"].values())
    many = f["at_1024_channels"]   # time only: blocks of 16384 there (design.choose_uniform_block)
    for key, parts in (("lowcut_44099_taps", 3), ("eq3_88197_taps", 6)):
        r = many[key]
        assert r["engine"] == "UpolsFirEngine" and r["block"] == 16384 and r["partitions"] == parts and r["us_per_call"] > 0 and r["roofline_frac"] > 0.05


def test_bench_exits_non_zero_without_a_line_when_the_world_is_not_what_was_asked_for():
    """a job whose process group is smaller than --gpus N (or whose ranks ended up with different filters) must not print
    a line the driver could take for an N-GPU measurement.  Simulated on one GPU: a world of one that was told to expect two."""
    out = _bench("--steps", "2", "--warmup", "1", "--no-stream-extra", "--no-latency", "--no-cpu-baseline", "--no-configs",
                 env={"ADSP_BENCH_FORCE_PG": "1", "MASTER_PORT": "29547", "ADSP_BENCH_EXPECT_RANKS": "2", "ADSP_BENCH_ABI_CHECK": "0"}, check=False)
    assert out.returncode == 3 and "ranks_seen = 1 (expected 2)" in out.stderr
    assert not any(ln.lstrip().startswith("{") for ln in out.stdout.splitlines())


# ---------------------------------------------------------------------------------------------------------------------
# 4. Live-session guards and the inspectable attributes of the drop-in classes
# ---------------------------------------------------------------------------------------------------------------------
def test_live_session_refuses_ring_and_configuration_calls(adsp):
    import ctypes
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design, _capi
    n = 512
    fir = FirStream(design.lowcut_kernel(300, 44100, n), n)
    eng = FirEngine(fir, channels=8, ring_slots=12)
    out = torch.empty((4, 8, n), device="cuda")
    eng.live_configure(step_timeout_ms=2000.0)
    eng.live_start(out, 4, 100, None)
    lib = _capi.load()
    p = ctypes.c_void_p(None)
    for call in (lambda: lib.adsp_ring_produce_begin(eng._h, ctypes.byref(p), None), lambda: lib.adsp_ring_produce_end(eng._h, None),
                 lambda: lib.adsp_apply_ring_resident(eng._h, ctypes.c_void_p(out.data_ptr()), 1, None), lambda: lib.adsp_set_accumulate(eng._h, 1),
                 lambda: lib.adsp_set_epilogue(eng._h, 1, 0.5, 0.0, 0.0), lambda: lib.adsp_set_block_outputs(eng._h, n)):
        assert call() == _capi.ADSP_ERR_STATE and b"live session" in lib.adsp_last_error()
    assert eng.live_stop() == 0
    eng.close()


def test_filtered_signal_and_original_signal_mirror_the_reference(adsp):
    """EffectFFTFilter.py:39 / :67-73 and EffectEQ3BandFFT.py:147-148 / :175-176."""
    n = 512
    adsp.config.initialize(44100, n)
    lc, eq = adsp.CreateLowCutFilter(200), adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5)
    assert lc.filtered_signal.shape == (3 * n,) and not lc.filtered_signal.any()
    assert eq.filtered_signal.shape == (3 * n,) and eq.original_signal.shape == (3 * n,) and not eq.original_signal.any()
    x = seeded_stream(3, 2 * n)
    y0, y1 = lc.apply(x[:n]), lc.apply(x[n:])
    assert lc.filtered_signal.dtype == np.complex128 and np.array_equal(lc.filtered_signal.real.astype(np.float32), y1) and y0.shape == (n,)
    eq.apply(x[:n])
    eq.apply(x[n:])
    assert np.array_equal(eq.original_signal, np.concatenate([np.zeros(n), x[:n], x[n:]])) and not eq.filtered_signal.any()


# ---------------------------------------------------------------------------------------------------------------------
# 5. Ring steps riding a live session (adsp_ring_set_pipeline(engine, 3))
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,kind,channels,slots", [(512, "eq", 4096, 40), (512, "lowcut", 70, 9), (2048, "lowcut", 33, 6), (128, "highcut", 1000, 12)])
def test_ring_steps_ride_a_live_session_with_a_real_producer(adsp, n, kind, channels, slots):
    """the per-step entry points (adsp_ring_acquire_stream -> producer on the caller's stream -> adsp_apply_ring ->
    adsp_ring_join; the reference's call pattern Example3.py:20-24, EffectEQ3BandFFT.py:156-211 per chunk) at pipeline depth 3: every
    step names its own output buffer, a real copy fills each slot on the caller's stream, the outputs equal the float64 direct sum;
    an ordinary call in between winds the session down and the next step starts a new one; an idle session ends by its time-out and
    is restarted; the depth falls back where no session can hold the engine."""
    import ctypes
    import time
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design, _capi
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    fs = 44100
    taps = {"eq": lambda: design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n), "lowcut": lambda: design.lowcut_kernel(300, fs, n),
            "highcut": lambda: design.highcut_kernel(5000, fs, n)}[kind]()
    fir = FirStream(taps, n)
    steps = 3 * slots + 5   # several ring laps
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(n + channels))
    t = _exact(adsp, fir, x)
    scale = float(t.abs().max())
    eng = FirEngine(fir, channels=channels, ring_slots=slots, optimize_for="stream")
    eng.live_configure(step_timeout_ms=300.0)
    assert eng.ring_set_pipeline("auto") == 3
    y = torch.full_like(x, 7.0)
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    user = torch.cuda.Stream()
    sp = user.cuda_stream
    plane = channels * n * 4

    def feed(k0, k1):
        for k in range(k0, k1):
            slot = eng.ring_acquire(sp)
            assert hip.hipMemcpyAsync(slot, x[k].data_ptr(), plane, 3, sp) == 0   # the producer: a device copy on the caller's stream
            eng.apply_ring(y[k], sp)
        eng.ring_join(sp)
    a, b = steps // 3, 2 * steps // 3
    feed(0, a)
    assert float((y[:a] - t[:a]).abs().max()) <= 1e-5 * scale and float(y[a:].min()) == 7.0
    # an ordinary launch in between: the session is wound down, the stream continues, the next ring step starts a new session
    eng.apply_device(x[a], y[a], 1, sp)
    user.synchronize()
    feed(a + 1, b)
    time.sleep(0.7)            # idle beyond the time-out: the session ends by itself ...
    feed(b, steps)             # ... and the next step starts a fresh one on the same history
    user.synchronize()
    assert float((y - t).abs().max()) <= 1e-5 * scale
    assert eng.ring_set_pipeline(1) == 1   # (winds the session down)
    eng.close()


def test_ring_pipeline_depth_3_is_refused_where_no_session_fits(adsp):
    from pyaudiodsptools_amd import FirEngine, FirStream, design, _capi
    n = 4096
    fir = FirStream(design.lowcut_kernel(800, 44100, n), n)
    eng = FirEngine(fir, channels=4096, ring_slots=4, optimize_for="stream")   # config 2: 4096 workgroups of 256 threads are not co-resident
    with pytest.raises(_capi.AdspError):
        eng.ring_set_pipeline(3)
    assert eng.ring_set_pipeline("auto") == 2
    eng.close()
    eng = FirEngine(fir, channels=8, ring_slots=3, optimize_for="stream")      # history + 1 slots: no pipelining of any depth
    with pytest.raises(_capi.AdspError):
        eng.ring_set_pipeline(3)
    eng.close()


def test_filter_change_in_a_stream_whose_steps_ride_a_session(adsp):
    """adsp_set_spectrum_async between two ring steps at pipeline depth 3: the session is wound down, the tables are updated on the caller's
    stream, the next step starts a new session BEHIND that update (the session runs on a stream of its own).  A streaming FIR's output
    depends on the input history only, so the steps after the change equal the new filter applied to the whole stream."""
    import ctypes
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    n, channels, steps, cut = 512, 300, 24, 11
    fir_a = FirStream(design.lowcut_kernel(300, 44100, n), n)
    fir_b = FirStream(design.highcut_kernel(3000, 44100, n), n)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(77))
    ta, tb = _exact(adsp, fir_a, x), _exact(adsp, fir_b, x)
    eng = FirEngine(fir_a, channels=channels, ring_slots=8, optimize_for="stream")
    assert eng.ring_set_pipeline("auto") == 3
    y = torch.empty_like(x)
    torch.cuda.synchronize()   # (x, ta, tb were produced on torch's default stream; the session and `user` are streams of their own)
    user = torch.cuda.Stream()
    sp = user.cuda_stream
    for k in range(steps):
        if k == cut:  # (no join before it: winding the session down consumes every submitted step first)
            eng.set_fir(fir_b, stream=sp, live=True)
        slot = eng.ring_acquire(sp)
        assert hip.hipMemcpyAsync(slot, x[k].data_ptr(), channels * n * 4, 3, sp) == 0
        eng.apply_ring(y[k], sp)
    eng.ring_join(sp)
    user.synchronize()
    scale = float(max(ta.abs().max(), tb.abs().max()))
    assert float((y[:cut] - ta[:cut]).abs().max()) <= 1e-5 * scale and float((y[cut:] - tb[cut:]).abs().max()) <= 1e-5 * scale
    eng.close()


@pytest.mark.parametrize("carry", [-1, 1])
@pytest.mark.parametrize("seed", range(10))
def test_upols_engine_randomised_shapes(adsp, seed, carry):
    """Seeded random long-kernel streams (chunk sizes that are multiples of 4, kernels of 1.1 - 3 chunks, any delay, 1 - 40 channels, calls
    of 1 - 3 chunks with sub-call splitting, float32 / int16): every sample against the float64 direct sum on the GPU."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(2500, 10000)) * 4
    taps_len = int(rng.integers(max(33000, int(1.1 * n)), 3 * n))
    latency = int(rng.integers(1, 4))
    lookahead = int(rng.integers(0, max(1, latency * n - 8192 - 4)))   # delay = latency * n - lookahead >= a block
    channels = int(rng.integers(1, 41))
    calls = [int(c) for c in rng.integers(1, 4, size=int(rng.integers(2, 5)))]
    fmt = "s16" if seed % 4 == 3 else "f32"
    taps = rng.standard_normal(taps_len) * np.hanning(taps_len) / np.sqrt(taps_len) * 2.0
    fir = adsp.FirStream(taps, n, latency_chunks=latency, lookahead=lookahead)
    eng = adsp.UpolsFirEngine(fir, channels=channels, sample_format=fmt, max_steps=int(rng.integers(1, 3)), block=8192 if seed % 2 or fir.delay - fir.delay % 4 < 16384 else 16384)
    eng.set_carry(carry)   # (1: the block that straddles a call boundary is always carried over - with 1 - 40 channels the library would not)
    steps = sum(calls)
    g = torch.Generator(device="cuda").manual_seed(seed)
    if fmt == "s16":
        x = torch.randint(-9000, 9000, (steps, channels, n), device="cuda", dtype=torch.int16, generator=g)
    else:
        x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=g)
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    pos = 0
    for k in calls:
        eng.apply_device(x[pos:pos + k], y[pos:pos + k], k, s)
        pos += k
    torch.cuda.synchronize()
    t = _exact(adsp, fir, x, fmt)
    what = f"seed {seed}: N={n} taps={taps_len} latency={latency} lookahead={lookahead} C={channels} calls={calls} {fmt} B={eng.block}"
    if fmt == "s16":
        diff = (y.int() - t.int()).abs()
        assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) <= 0.01, what
    else:
        scale = float(t.abs().max())
        assert scale > 0.05 and float((y - t).abs().max()) <= 1e-5 * scale, what
    eng.close()


def test_graft_entry_smoke_runs():
    """The driver's smoke(): one small invocation of the hot path (and of a long kernel) on cuda:0 against the oracle."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.parametrize("n", [4096, 88200])
def test_gpu_twins_keep_a_device_resident_chunk_on_the_gpu(adsp, n):
    """Example4.py:9-21 / ModuleTestsGPU.py:58: the *GPU classes are fed device arrays (cupy there, torch tensors here) and return device
    arrays; the stream must be the one the numpy call produces, and the inspectable attributes still read back."""
    import torch
    adsp.config.initialize(44100, n)
    x = seeded_stream(n, 4 * n)
    for make in (lambda: adsp.CreateLowCutFilterGPU(800), lambda: adsp.CreateEQ3BandFFTGPU(100, 2, 700, -4, 8000, 5)):
        host_dev, gpu_dev = make(), make()
        want = np.concatenate([host_dev.apply(x[i * n:(i + 1) * n]) for i in range(4)])
        xd = torch.from_numpy(x).cuda()
        outs = [gpu_dev.apply(xd[i * n:(i + 1) * n]) for i in range(4)]
        assert all(isinstance(o, torch.Tensor) and o.is_cuda and o.dtype == torch.float32 and o.shape == (n,) for o in outs)
        got = torch.cat(outs).cpu().numpy()
        assert np.array_equal(got, want)                      # the same kernels on the same data: bit for bit
        assert np.array_equal(gpu_dev.original_signal, host_dev.original_signal) if hasattr(gpu_dev, "original_signal") else True
        assert np.asarray(gpu_dev.filtered_signal).shape == np.asarray(host_dev.filtered_signal).shape
        with pytest.raises(ValueError):
            gpu_dev.apply(xd[:n - 4])
    adsp.config.initialize(44100, 4096)
