"""GPU parity: the HIP path (through the C ABI) against golden vectors from the reference, the CPU
oracle, and size-independent properties at BASELINE.json's full sizes.  Run with -m gpu on MI355X."""
import numpy as np
import pytest

from conftest import assert_parity, seeded_stream

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


KAT = {
    # name: (factory(adsp), fs, N, seed, chunks)
    "A": (lambda p: p.CreateLowCutFilter(800), 44100, 4096, 1234, 6),
    "B": (lambda p: p.CreateHighCutFilter(8000), 44100, 4096, 1234, 6),
    "C": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 44100, 512, 1234, 6),
    "D": (lambda p: p.CreateLowCutFilter(200), 44100, 512, 1234, 6),
    "EQ4096": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 44100, 4096, 77, 5),
    "LC8192": (lambda p: p.CreateLowCutFilter(800), 96000, 8192, 78, 4),
    "EQ8192": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 96000, 8192, 79, 4),
    "HC1024": (lambda p: p.CreateHighCutFilter(20000), 48000, 1024, 80, 7),
    "EQ1024": (lambda p: p.CreateEQ3BandFFT(250, -6, 1500, 3, 6000, -2.5), 48000, 1024, 81, 7),
    "LC2048": (lambda p: p.CreateLowCutFilter(160), 44100, 2048, 82, 5),
    "HC256": (lambda p: p.CreateHighCutFilter(3000), 44100, 256, 83, 9),
    "EQ128": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 44100, 128, 84, 9),
    "LC64": (lambda p: p.CreateLowCutFilter(2000), 44100, 64, 85, 11),
    # chunk sizes that are not powers of two run on the generic-geometry kernel
    "LC1000": (lambda p: p.CreateLowCutFilter(300), 44100, 1000, 86, 7),
    "EQ1000": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 44100, 1000, 87, 7),
    "HC1920": (lambda p: p.CreateHighCutFilter(9000), 48000, 1920, 88, 5),
    "LC12000": (lambda p: p.CreateLowCutFilter(120), 44100, 12000, 89, 3),
    "EQ20": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 44100, 20, 90, 40),
    # chunk sizes that are NOT multiples of 4 (round 4): the kernel's dword-access form; N // 2 odd gives an even
    # filter length, whose slice look-ahead is L // 2 (EffectFFTFilter.py:22-25)
    "LC30": (lambda p: p.CreateLowCutFilter(3000), 44100, 30, 93, 30),
    "HC30": (lambda p: p.CreateHighCutFilter(8000), 44100, 30, 94, 30),
    "EQ30": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 44100, 30, 95, 30),
    "LC1001": (lambda p: p.CreateLowCutFilter(300), 44100, 1001, 96, 7),
    "EQ1001": (lambda p: p.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 44100, 1001, 97, 7),
    "LC1002": (lambda p: p.CreateLowCutFilter(500), 48000, 1002, 98, 7),
    "HC1002": (lambda p: p.CreateHighCutFilter(9000), 48000, 1002, 99, 7),
    "EQ1002": (lambda p: p.CreateEQ3BandFFT(250, -6, 1500, 3, 6000, -2.5), 48000, 1002, 100, 7),
    "HC6": (lambda p: p.CreateHighCutFilter(8000), 44100, 6, 101, 50),
    "LC4410": (lambda p: p.CreateLowCutFilter(160), 44100, 4410, 102, 4),
}


@pytest.mark.parametrize("name", sorted(KAT))
def test_dropin_apply_matches_reference_golden(adsp, golden, name):
    """The reference's own call pattern: config.initialize, Create*, one .apply per chunk."""
    make, fs, n, seed, chunks = KAT[name]
    adsp.config.initialize(fs, n)
    dev = make(adsp)
    x = seeded_stream(seed, chunks * n)
    outs = [dev.apply(x[i * n:(i + 1) * n]) for i in range(chunks)]
    assert all(o.dtype == np.float32 and o.shape == (n,) for o in outs)
    assert_parity(np.concatenate(outs), golden["kat_streams"][name], what=name)


@pytest.mark.parametrize("name", sorted(KAT))
def test_exact_mode_matches_reference_golden_to_float32_rounding(adsp, golden, name):
    """The on-device ground truth of the full-size tests (ExactFirEngine, float64 direct sum) against every stream captured
    from the reference: they agree to the reference's own float32/complex64 rounding (a few 1e-7 of full scale) - thirty
    times closer than the 1e-5 budget the FFT engines are held to."""
    make, fs, n, seed, chunks = KAT[name]
    adsp.config.initialize(fs, n)
    fir = make(adsp).fir
    x = seeded_stream(seed, chunks * n).reshape(chunks, 1, n)
    y = adsp.ExactFirEngine(fir, channels=1).apply_host(x).reshape(-1)
    ref = golden["kat_streams"][name]
    assert np.abs(y - ref).max() <= 2e-7 * max(np.abs(ref).max(), 1.0), np.abs(y - ref).max()  # measured: 4e-8 .. 9e-8


def test_default_arguments_and_attributes(adsp, golden):
    adsp.config.initialize(44100, 512)
    hc, lc = adsp.CreateHighCutFilter(), adsp.CreateLowCutFilter()
    assert (hc.fH, lc.fH, hc.fS, hc.filter_length) == (8000, 160, 44100, 255)
    assert (hc.array_slice_value_start, hc.array_slice_value_end) == (512 + 127, 512 - 127)
    assert np.abs(np.fft.ifft(hc.sinc_filter).real[:255] - golden["design"]["highcut_default_44100_512"]).max() < 1e-12
    assert np.abs(np.fft.ifft(lc.sinc_filter).real[:255] - golden["design"]["lowcut_default_44100_512"]).max() < 1e-12
    with pytest.raises(TypeError):
        adsp.CreateEQ3BandFFT()  # six positionals, no defaults (EffectEQ3BandFFT.py:47)


def test_config_is_read_at_construction_only(adsp, golden):
    adsp.config.initialize(44100, 512)
    dev = adsp.CreateLowCutFilter(200)
    adsp.config.initialize(48000, 1024)  # must not affect the existing device
    x = seeded_stream(1234, 6 * 512)
    y = np.concatenate([dev.apply(x[i * 512:(i + 1) * 512]) for i in range(6)])
    assert_parity(y, golden["kat_streams"]["D"], what="D after re-initialize")


EDGES = {
    "zeros": lambda n: np.zeros(5 * n, np.float32),
    "imp0": lambda n: np.eye(1, 5 * n, 0, dtype=np.float32)[0],
    "impNm1": lambda n: np.eye(1, 5 * n, n - 1, dtype=np.float32)[0],
    "impN": lambda n: np.eye(1, 5 * n, n, dtype=np.float32)[0],
    "dc": lambda n: np.ones(5 * n, np.float32),
    "square": lambda n: np.where((np.arange(5 * n) // 37) % 2 == 0, 1.0, -1.0).astype(np.float32),
}


@pytest.mark.parametrize("edge", sorted(EDGES))
def test_edge_inputs_G(adsp, golden, edge):
    n = 512
    x = EDGES[edge](n)
    adsp.config.initialize(44100, n)
    for tag, dev in (("lowcut_", adsp.CreateLowCutFilter(200)), ("highcut_", adsp.CreateHighCutFilter(8000)),
                     ("eq_", adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5))):
        y = np.concatenate([dev.apply(x[i * n:(i + 1) * n]) for i in range(5)])
        assert_parity(y, golden["kat_edges"][tag + edge], what=tag + edge)
    if edge == "zeros":
        assert not y.any()


def test_input_types_and_wrong_length(adsp, golden):
    n = 512
    adsp.config.initialize(44100, n)
    x64 = np.random.default_rng(5).uniform(-1, 1, 4 * n)
    dev = adsp.CreateLowCutFilter(200)
    y = np.concatenate([dev.apply(x64[i * n:(i + 1) * n]) for i in range(4)])
    assert_parity(y, golden["kat_edges"]["lowcut_f64in"], what="float64 input")
    dev = adsp.CreateHighCutFilter(8000)
    y = np.concatenate([dev.apply(list(x64[i * n:(i + 1) * n])) for i in range(4)])
    assert_parity(y, golden["kat_edges"]["highcut_listin"], what="list input")
    # wrong length: ValueError like the reference, but the history must be untouched
    dev = adsp.CreateHighCutFilter(8000)
    first = dev.apply(x64[:n])
    with pytest.raises(ValueError):
        dev.apply(x64[: n - 1])
    rest = [dev.apply(list(x64[i * n:(i + 1) * n])) for i in range(1, 4)]
    assert_parity(np.concatenate([first] + rest), golden["kat_edges"]["highcut_listin"], what="after bad call")


def test_example1_plumbing_F(adsp, golden):
    g = golden["kat_example1"]
    x = g["pcm16_first8"].astype(np.float32) / 32768
    adsp.config.initialize(44100, 4096)
    dev = adsp.CreateLowCutFilter(800)
    y = np.concatenate([dev.apply(x[i * 4096:(i + 1) * 4096]) for i in range(8)])
    assert_parity(y, g["out_first8"], what="Example1")


def test_chain_fused_E(adsp, golden):
    """config 5's chain as ONE engine == three reference devices in series (golden E)."""
    n, fs = 8192, 96000
    adsp.config.initialize(fs, n)
    eng = adsp.fuse(adsp.CreateLowCutFilter(800), adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5),
                    adsp.CreateHighCutFilter(8000))
    # the composite's 6976 negligible end taps are left out by default (FirStream.trimmed): 9401 of 16377 taps, so a
    # 4N transform keeps 2.75 N samples and the window reaches 4 chunks back
    assert eng.geometry.fft_size == 4 * n and eng.geometry.history_chunks == 4 and len(eng.fir.taps) == 9401
    assert eng.geometry.max_block_outputs == 11 * n // 4
    x = seeded_stream(4321, 12 * n).reshape(12, 1, n)
    y = np.concatenate([eng.apply_host(x[k])[0] for k in range(12)])
    assert_parity(y, golden["kat_chain"]["E"], what="chain streaming")
    eng.reset()
    y2 = eng.apply_host(x).reshape(-1)  # all 12 steps in one launch (multi-step blocks)
    assert_parity(y2, golden["kat_chain"]["E"], what="chain offline")
    # every tap kept (trim=0): the round-1 geometry, 2 N kept per 4N transform
    full = adsp.fuse(adsp.CreateLowCutFilter(800), adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5),
                     adsp.CreateHighCutFilter(8000), trim=0)
    assert full.geometry.history_chunks == 5 and len(full.fir.taps) == 16377 and full.geometry.max_block_outputs == 2 * n
    y3 = full.apply_host(x).reshape(-1)
    assert_parity(y3, golden["kat_chain"]["E"], what="chain offline, untrimmed")
    # what trimming costs: far below the float32 pipeline's own rounding
    assert np.abs(y3 - y2).max() <= 2e-6 * np.abs(y3).max()


@pytest.mark.parametrize("n,channels", [(64, 37), (128, 9), (256, 5), (512, 7), (1024, 3), (2048, 2), (4096, 5), (8192, 2)])
def test_batched_channels_vs_oracle(adsp, n, channels):
    """Ragged channel counts (not a multiple of channels-per-workgroup), every channel checked."""
    o = orc()
    fs, steps = 44100, 5
    adsp.config.initialize(fs, n)
    rng = np.random.default_rng(n + channels)
    x = rng.uniform(-1, 1, (steps, channels, n)).astype(np.float32)
    for make_dev, taps in (
        (lambda: adsp.CreateLowCutFilter(300, channels=channels), o.lowcut_taps(300, fs, n)),
        (lambda: adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5, channels=channels),
         o.eq3_composite_taps(100, 2, 700, -4, 8000, 5, fs, n)),
    ):
        dev = make_dev()
        y = np.stack([dev.apply_batch(x[k]) for k in range(steps)])
        for c in range(channels):
            truth = o.direct_stream_convolution(taps, x[:, c].reshape(-1), n)
            assert_parity(y[:, c].reshape(-1), truth, what=f"N={n} ch={c}")


@pytest.mark.parametrize("n", [512, 4096])
def test_multistep_equals_streaming_and_state_roundtrip(adsp, n):
    fs, channels, steps = 44100, 4, 9
    adsp.config.initialize(fs, n)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (steps, channels, n)).astype(np.float32)
    a = adsp.CreateLowCutFilter(800, channels=channels)
    stream = np.stack([a.apply_batch(x[k]) for k in range(steps)])
    b = adsp.CreateLowCutFilter(800, channels=channels)
    assert b.engine.block_outputs == n + n // 2  # 1.5 N kept per 2N transform in multi-step launches
    part1 = b.apply_batch(x[:4])
    state = b.engine.get_state()
    assert np.array_equal(state, x[2:4])  # the two previous chunks, oldest first
    part2 = b.apply_batch(x[4:])
    assert_parity(np.concatenate([part1, part2]), stream, what="multi-step vs streaming")
    # restore the saved state into a fresh device and continue from step 4
    c = adsp.CreateLowCutFilter(800, channels=channels)
    c.engine.set_state(state)
    assert_parity(c.apply_batch(x[4:]), part2, what="state round trip")
    c.reset()
    assert not c.engine.get_state().any()
    assert_parity(c.apply_batch(x[:4]), part1, what="after reset")
    # block size N as well (different transform count, same samples)
    b.reset()
    b.engine.set_block_outputs(n)
    assert_parity(b.apply_batch(x), stream, what="V=N multi-step")


@pytest.mark.parametrize("n", [512, 4096])
def test_batch_optimised_geometry_eq(adsp, n):
    """optimize_for="batch" runs the EQ on a 4N transform (3N kept): same stream as the 2N engine and the oracle."""
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    o = orc()
    fs, channels, steps = 44100, 3, 7
    taps = design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n)
    fir = FirStream(taps, n)
    eng = FirEngine(fir, channels=channels, optimize_for="batch")
    assert eng.geometry.fft_size == 4 * n and eng.block_outputs == 3 * n
    x = np.random.default_rng(n).uniform(-1, 1, (steps, channels, n)).astype(np.float32)
    y = np.concatenate([eng.apply_host(x[:4]), eng.apply_host(x[4:5]), eng.apply_host(x[5:])])  # multi-step, single, multi
    for c in range(channels):
        assert_parity(y[:, c].reshape(-1), o.direct_stream_convolution(taps, x[:, c].reshape(-1), n), what=f"ch {c}")


@pytest.mark.parametrize("n,channels", [(20, 37), (1000, 5), (1920, 3), (3000, 4), (12000, 2), (16384, 2), (44100, 1)])
def test_generic_chunk_sizes_vs_oracle(adsp, n, channels):
    """Any chunk size divisible by 4: batched channels, per-step and multi-step launches, float32 and int16."""
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    o = orc()
    fs, steps = 44100, 5
    taps = design.lowcut_kernel(250, fs, n)
    fir = FirStream(taps, n)
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, (steps, channels, n)).astype(np.float32)
    truth = [o.direct_stream_convolution(taps, x[:, c].reshape(-1), n) for c in range(channels)]
    for mode in ("stream", "batch"):
        eng = FirEngine(fir, channels=channels, optimize_for=mode)
        y_stream = np.stack([eng.apply_host(x[k]) for k in range(steps)])
        eng.reset()
        y_multi = np.concatenate([eng.apply_host(x[:2]), eng.apply_host(x[2:])])
        for c in range(channels):
            assert_parity(y_stream[:, c].reshape(-1), truth[c], what=f"N={n} {mode} stream ch {c}")
            assert_parity(y_multi[:, c].reshape(-1), truth[c], what=f"N={n} {mode} multi ch {c}")
    pcm = rng.integers(-8000, 8000, (steps, channels, n), dtype=np.int16)
    eng = FirEngine(fir, channels=channels, sample_format="s16")
    y = eng.apply_host(pcm)
    for c in range(channels):
        want = o.float_to_pcm16(o.direct_stream_convolution(taps, o.pcm16_to_float(pcm[:, c].reshape(-1)), n).astype(np.float32))
        assert np.abs(y[:, c].reshape(-1).astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_example4_chunk_size_partitioned(adsp, golden):
    """N = 88200 (Example4.py:5): 44099 / 88197 taps do not fit one transform.  Round 5: the drop-in classes run the uniformly
    partitioned engine (one forward transform per input block, frequency-domain delay line, one inverse per output block);
    PartitionedFirEngine (one engine pass per kernel slice, summed) stays as the checker.  Both against the reference's decimated
    goldens, host path and device path."""
    import torch
    n = 88200
    adsp.config.initialize(44100, n)
    for dev, seed, key in ((adsp.CreateLowCutFilter(300), 91, "LC88200_dec64"),
                           (adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), 92, "EQ88200_dec64")):
        assert type(dev.engine).__name__ == "UpolsFirEngine" and dev.engine.partition.n_partitions in (6, 11)
        x = seeded_stream(seed, 3 * n)
        y = np.concatenate([dev.apply(x[i * n:(i + 1) * n]) for i in range(3)])
        assert_parity(y[::64], golden["kat_streams"][key], what=key)
        for eng in (dev.engine, adsp.PartitionedFirEngine(dev.fir)):
            assert type(eng).__name__ != "PartitionedFirEngine" or len(eng.engines) >= 4
            eng.reset()
            xd = torch.from_numpy(x.reshape(3, 1, n)).cuda()
            yd = torch.full_like(xd, 7.0)  # every sample must be overwritten (partitioned: part 0 overwrites, the others accumulate)
            torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
            s = torch.cuda.current_stream().cuda_stream
            eng.apply_device(xd[:2], yd[:2], 2, s)
            eng.apply_device(xd[2], yd[2], 1, s)
            torch.cuda.synchronize()
            assert_parity(yd.cpu().numpy().reshape(-1), y, what=key + " device path " + type(eng).__name__)
    adsp.config.initialize(44100, 88200)
    with pytest.raises(ValueError):
        adsp.FirEngine(dev.fir, sample_format="s16")  # one engine cannot hold it ...
    assert type(adsp.make_engine(dev.fir, sample_format="s16")).__name__ == "UpolsFirEngine"  # ... the partitioned one can (round 5)


def test_device_pointer_and_ring_paths(adsp):
    """adsp_apply_device on torch tensors and the zero-copy ring path give the same stream."""
    import torch
    n, channels, steps = 1024, 6, 7
    adsp.config.initialize(48000, n)
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    fir = FirStream(design.highcut_kernel(5000, 48000, n), n)
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (steps, channels, n)).astype(np.float32)
    ref_eng = FirEngine(fir, channels=channels)
    ref = ref_eng.apply_host(x)
    xd = torch.from_numpy(x).cuda()
    s = torch.cuda.current_stream().cuda_stream
    # per-step device calls
    e1 = FirEngine(fir, channels=channels)
    yd = torch.empty_like(xd)
    for k in range(steps):
        e1.apply_device(xd[k], yd[k], 1, s)
    torch.cuda.synchronize()
    assert_parity(yd.cpu().numpy(), ref, what="apply_device per step")
    # zero-copy ring: the producer writes into the slot the engine hands out
    e2 = FirEngine(fir, channels=channels, ring_slots=5)
    y2 = torch.empty_like(xd)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    for k in range(steps):
        slot = e2.ring_acquire()
        assert hip.hipMemcpyAsync(slot, xd[k].data_ptr(), channels * n * 4, 3, s) == 0
        e2.apply_ring(y2[k], s)
    torch.cuda.synchronize()
    assert_parity(y2.cpu().numpy(), ref, what="ring path")


@pytest.mark.parametrize("live", [False, True])
def test_spectrum_update_keeps_history(adsp, live):
    """Filter change between two steps: adsp_set_spectrum (drains the device) and adsp_set_spectrum_async (stream-ordered,
    no synchronisation: steps queued before it use the old filter, steps queued after it the new one)."""
    import torch
    n, channels = 512, 3
    adsp.config.initialize(44100, n)
    from pyaudiodsptools_amd import FirStream, design
    o = orc()
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (6, channels, n)).astype(np.float32)
    dev = adsp.CreateLowCutFilter(200, channels=channels)
    new_taps = design.lowcut_kernel(1000, 44100, n)
    old_taps = design.lowcut_kernel(200, 44100, n)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    s = torch.cuda.current_stream().cuda_stream
    for k in range(3):
        dev.engine.apply_device(xd[k], yd[k], 1, s)
    dev.engine.set_fir(FirStream(new_taps, n), stream=s, live=live)  # no host synchronisation in between when live
    for k in range(3, 6):
        dev.engine.apply_device(xd[k], yd[k], 1, s)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    before = np.stack([o.direct_stream_convolution(old_taps, x[:, c].reshape(-1), n)[:3 * n] for c in range(channels)])
    after = np.stack([o.direct_stream_convolution(new_taps, x[:, c].reshape(-1), n)[3 * n:] for c in range(channels)])
    assert_parity(y[:3].transpose(1, 0, 2).reshape(channels, -1), before, what="before the filter change")
    assert_parity(y[3:].transpose(1, 0, 2).reshape(channels, -1), after, what="after set_fir")


def test_errors_cross_the_abi_as_exceptions(adsp):
    from pyaudiodsptools_amd import _capi
    adsp.config.initialize(44100, 3)
    with pytest.raises(ValueError):
        adsp.CreateLowCutFilter(800)  # N // 2 - 1 = 0 taps: no filter (round 4: every chunk size from 4 up is accepted, 3002 included)
    adsp.config.initialize(44100, 3002)
    assert adsp.CreateLowCutFilter(800).apply(np.zeros(3002, np.float32)).shape == (3002,)
    adsp.config.initialize(44100, 512)
    dev = adsp.CreateLowCutFilter(800, channels=2)
    with pytest.raises(ValueError):
        dev.apply(np.zeros(512, np.float32))  # multi-channel device needs apply_batch
    with pytest.raises(ValueError):
        dev.apply_batch(np.zeros((3, 512), np.float32))
    with pytest.raises(_capi.AdspError):
        dev.engine.set_block_outputs(100)


# ---- BASELINE.json full sizes: properties that do not need the oracle at full size --------------
def oracle_channels(channels, first=2, seed=0):
    """>= 32 channels for the oracle's float64 direct convolution: the first and the last workgroups of the grid, every
    residue c % 8 (the XCD a channel group lands on) in both, and sixteen more spread over the rest."""
    rng = np.random.default_rng(seed)
    head = list(range(first, first + 8))
    tail = list(range(channels - 8, channels))
    mid = sorted(int(c) for c in rng.choice(np.arange(first + 8, channels - 8), 16, replace=False))
    mid = [c + ((i - c) % 8) for i, c in enumerate(mid)]  # residues 0..7 twice
    out = sorted(set(head + mid + tail))
    assert len(out) >= 32 and {c % 8 for c in out} == set(range(8)) and out[-1] == channels - 1
    return out


def assert_all_channels_match_exact(adsp, fir, x, y, what):
    """EVERY output sample of a full-size batch against the float64 direct sum computed on the GPU (adsp_exact_*, itself
    pinned to the oracle's direct_stream_convolution by tests/test_gpu_pcm16.py): the 1e-5 tolerance of BASELINE.md 3."""
    import torch
    ex = adsp.ExactFirEngine(fir, channels=x.shape[1])
    truth = torch.empty_like(x)
    ex.apply_device(x, truth, x.shape[0], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    scale = float(truth.abs().max())
    err = float((y - truth).abs().max())
    assert scale > 0.1 and err <= 1e-5 * scale, f"{what}: max|d| = {err:.3e} over all channels, scale {scale:.3e}"
    assert torch.allclose(y, truth, rtol=1e-5, atol=1e-6 * max(scale, 1.0)), f"{what}: allclose failed"
    ex.close()



def test_config2_full_size_properties(adsp):
    """4096 channels x 4096 samples, LowCut(800): impulse response == taps, linearity, and a sampled
    oracle check, through the device-pointer path."""
    import torch
    o = orc()
    n, channels, steps, fs = 4096, 4096, 4, 44100
    adsp.config.initialize(fs, n)
    dev = adsp.CreateLowCutFilter(800, channels=channels)
    eng = dev.engine
    taps = o.lowcut_taps(800, fs, n)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=g)
    x[:, 0] = 0
    x[0, 0, 17] = 1.0  # channel 0: unit impulse at sample 17
    x[:, 1] = 0        # channel 1: silence
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for k in range(steps):
        eng.apply_device(x[k], y[k], 1, s)
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    assert np.isfinite(yh).all()
    assert not yh[:, 1].any()
    d = n // 4 - 1
    imp = yh[:, 0].reshape(-1)
    start = n - d + 17  # out[tau] = sum c[t] s[tau - N + d - t]
    assert np.abs(imp[start:start + len(taps)] - taps).max() <= 1e-5 * np.abs(taps).max()
    assert np.abs(np.delete(imp, np.arange(start, start + len(taps)))).max() <= 2e-6
    xh = x.cpu().numpy()
    for c in oracle_channels(channels):
        truth = o.direct_stream_convolution(taps, xh[:, c].reshape(-1), n)
        assert_parity(yh[:, c].reshape(-1), truth, what=f"config2 ch {c}")
    assert_all_channels_match_exact(adsp, dev.fir, x, y, "config2")
    # linearity / channel independence: filter(a*x + b*z) == a*filter(x) + b*filter(z), in one multi-step launch
    eng.reset()
    z = torch.empty_like(x).uniform_(-1, 1, generator=g)
    mix = 0.5 * x + 0.25 * z
    y_mix, y_z = torch.empty_like(x), torch.empty_like(x)
    eng.apply_device(mix, y_mix, steps, s)
    eng.reset()
    eng.apply_device(z, y_z, steps, s)
    torch.cuda.synchronize()
    lin = 0.5 * y + 0.25 * y_z
    assert float((y_mix - lin).abs().max()) <= 1e-5 * float(lin.abs().max())


def test_config3_full_size_eq_stereo_pairs(adsp):
    """2048 stereo pairs x 512 samples, EQ3: L and R of a pair fed the same signal give the same output;
    sampled channels match the oracle."""
    import torch
    o = orc()
    n, channels, steps, fs = 512, 4096, 6, 44100
    adsp.config.initialize(fs, n)
    dev = adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5, channels=channels)
    g = torch.Generator(device="cuda").manual_seed(6)
    mono = torch.empty((steps, channels // 2, n), device="cuda").uniform_(-1, 1, generator=g)
    x = mono.repeat_interleave(2, dim=1).contiguous()
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for k in range(steps):
        dev.engine.apply_device(x[k], y[k], 1, s)
    torch.cuda.synchronize()
    assert torch.equal(y[:, 0::2], y[:, 1::2])
    taps = o.eq3_composite_taps(100, 2, 700, -4, 8000, 5, fs, n)
    xh, yh = x.cpu().numpy(), y.cpu().numpy()
    for c in oracle_channels(channels, first=0):
        assert_parity(yh[:, c].reshape(-1), o.direct_stream_convolution(taps, xh[:, c].reshape(-1), n), what=f"config3 ch {c}")


def test_config4_full_size_properties(adsp):
    """config 4's per-GPU shape: 8192 channels x 4096 samples, HighCut(8000) - impulse response == taps, silence,
    linearity, sampled channels against the float64 direct convolution; streaming and one multi-step launch."""
    import torch
    o = orc()
    n, channels, steps, fs = 4096, 8192, 3, 44100
    adsp.config.initialize(fs, n)
    dev = adsp.CreateHighCutFilter(8000, channels=channels)
    eng = dev.engine
    taps = o.highcut_taps(8000, fs, n)
    g = torch.Generator(device="cuda").manual_seed(15)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=g)
    x[:, 0] = 0
    x[0, 0, n - 1] = 1.0  # channel 0: unit impulse on the last sample of the first chunk
    x[:, 1] = 0           # channel 1: silence
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for k in range(steps):
        eng.apply_device(x[k], y[k], 1, s)
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    assert np.isfinite(yh).all() and not yh[:, 1].any()
    d = n // 4 - 1
    imp = yh[:, 0].reshape(-1)
    start = n - d + (n - 1)
    assert np.abs(imp[start:start + len(taps)] - taps).max() <= 1e-5 * np.abs(taps).max()
    assert np.abs(np.delete(imp, np.arange(start, start + len(taps)))).max() <= 2e-6
    xh = x.cpu().numpy()
    for c in oracle_channels(channels):
        assert_parity(yh[:, c].reshape(-1), o.direct_stream_convolution(taps, xh[:, c].reshape(-1), n), what=f"config4 ch {c}")
    assert_all_channels_match_exact(adsp, dev.fir, x, y, "config4")
    eng.reset()
    y_all = torch.empty_like(x)
    eng.apply_device(x, y_all, steps, s)  # one launch, 1.5 N kept per transform
    torch.cuda.synchronize()
    assert float((y_all - y).abs().max()) <= 2e-6
    eng.reset()
    z = torch.empty_like(x).uniform_(-1, 1, generator=g)
    y_mix, y_z = torch.empty_like(x), torch.empty_like(x)
    eng.apply_device(0.5 * x + 0.25 * z, y_mix, steps, s)
    eng.reset()
    eng.apply_device(z, y_z, steps, s)
    torch.cuda.synchronize()
    lin = 0.5 * y + 0.25 * y_z
    assert float((y_mix - lin).abs().max()) <= 1e-5 * float(lin.abs().max())


def test_config5_full_size_properties(adsp):
    """config 5's per-GPU shape: 4096 channels x 8192 samples @ 96 kHz, LowCut -> EQ3 -> HighCut as ONE engine - impulse
    response == the full 16377-tap composite, silence, linearity, sampled channels against the float64 direct
    convolution of the three devices in series; one multi-step launch (2.75 N kept per 4N transform) and streaming."""
    import torch
    from pyaudiodsptools_amd import design
    o = orc()
    n, channels, steps, fs = 8192, 4096, 11, 96000
    adsp.config.initialize(fs, n)
    eng = adsp.fuse(adsp.CreateLowCutFilter(800), adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5),
                    adsp.CreateHighCutFilter(8000), channels=channels, optimize_for="batch")
    chain = (design.FirStream(o.lowcut_taps(800, fs, n), n).then(design.FirStream(o.eq3_composite_taps(100, 2, 700, -4, 8000, 5, fs, n), n))
             .then(design.FirStream(o.highcut_taps(8000, fs, n), n)))
    taps = chain.taps  # every tap: the engine runs the trimmed kernel, the truth does not
    assert len(taps) == 16377 and chain.latency_chunks == 3
    g = torch.Generator(device="cuda").manual_seed(16)
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=g)
    x[:, 0] = 0
    x[1, 0, 5] = 1.0  # channel 0: unit impulse
    x[:, 1] = 0       # channel 1: silence
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    eng.apply_device(x, y, steps, s)
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    assert np.isfinite(yh).all() and not yh[:, 1].any()
    imp = yh[:, 0].reshape(-1)
    start = chain.delay + n + 5  # out[tau] = sum c[t] s[tau - delay - t]
    assert np.abs(imp[start:start + len(taps)] - taps).max() <= 1e-5 * np.abs(taps).max()
    assert np.abs(np.delete(imp, np.arange(start, start + len(taps)))).max() <= 2e-6
    xh = x.cpu().numpy()
    from scipy.signal import fftconvolve
    for i, c in enumerate(oracle_channels(channels)):
        if i % 8 == 0:  # four channels against the direct sum (16377 taps x 32768 samples: 6 s each) ...
            truth = o.direct_stream_convolution(taps, xh[:, c].reshape(-1), n, chain.latency_chunks, chain.lookahead)
        else:           # ... the other 28 against scipy's float64 FFT convolution of the same definition (1e-15)
            s64 = xh[:, c].reshape(-1).astype(np.float64)
            full = fftconvolve(s64, taps)
            truth = np.zeros(len(s64))
            truth[chain.delay:] = full[: len(s64) - chain.delay]
        assert_parity(yh[:, c].reshape(-1), truth, what=f"config5 ch {c}")
    assert_all_channels_match_exact(adsp, chain, x, y, "config5")  # against ALL 16377 taps, every channel
    # streaming (one launch per chunk) == the multi-step launch
    eng.reset()
    y_s = torch.empty_like(x)
    for k in range(steps):
        eng.apply_device(x[k], y_s[k], 1, s)
    torch.cuda.synchronize()
    assert float((y_s - y).abs().max()) <= 4e-6
    eng.reset()
    z = torch.empty_like(x).uniform_(-1, 1, generator=g)
    y_mix, y_z = torch.empty_like(x), torch.empty_like(x)
    eng.apply_device(0.5 * x + 0.25 * z, y_mix, steps, s)
    eng.reset()
    eng.apply_device(z, y_z, steps, s)
    torch.cuda.synchronize()
    lin = 0.5 * y + 0.25 * y_z
    assert float((y_mix - lin).abs().max()) <= 1e-5 * float(lin.abs().max())


def test_sharded_bank_rccl_world1_real_engine(adsp):
    """The N-GPU layer with the REAL HIP engine on the one GPU a test box has: init_process_group("nccl") (= RCCL),
    the spectrum broadcast as a device tensor, uploaded with adsp_set_spectrum_device from the tensor the collective
    left on the GPU.  World size 1 is all a one-GPU box can prove; the same code runs per rank on eight."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from pyaudiodsptools_amd import design
    from pyaudiodsptools_amd.dist import ShardedFirBank, init_process_group
    o = orc()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_process_group("nccl")
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        n, fs, channels, steps = 4096, 44100, 48, 4
        taps = o.lowcut_taps(800, fs, n)
        bank = ShardedFirBank(design.FirStream(design.lowcut_kernel(800, fs, n), n), channels, device=0, optimize_for="batch")
        assert (bank.lo, bank.hi) == (0, channels) and bank.spectrum_tensor.is_cuda
        assert bank.engine.real_spectrum  # the zero-phase spectrum survived broadcast + device upload bit for bit
        x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(17))
        y = torch.empty_like(x)
        bank.engine.apply_device(x, y, steps, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        xh, yh = x.cpu().numpy(), y.cpu().numpy()
        for c in (0, 17, channels - 1):
            assert_parity(yh[:, c].reshape(-1), o.direct_stream_convolution(taps, xh[:, c].reshape(-1), n), what=f"rccl world-1 ch {c}")
        # a second filter through the same path: complex spectrum (EQ), different geometry
        eq = design.FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, 512), 512)
        bank2 = ShardedFirBank(eq, 5, device=0)
        assert not bank2.engine.real_spectrum
        xe = np.random.default_rng(18).uniform(-1, 1, (6, 5, 512)).astype(np.float32)
        ye = np.stack([bank2.engine.apply_host(xe[k]) for k in range(6)])
        et = o.eq3_composite_taps(100, 2, 700, -4, 8000, 5, fs, 512)
        assert_parity(ye[:, 3].reshape(-1), o.direct_stream_convolution(et, xe[:, 3].reshape(-1), 512), what="rccl world-1 eq")
    finally:
        dist.destroy_process_group()


def test_raw_c_abi_error_paths(adsp):
    """Status codes + adsp_last_error() for misuse, straight through ctypes (no Python wrapper logic)."""
    import ctypes
    from pyaudiodsptools_amd import _capi
    lib = _capi.load()

    def create(**kw):
        base = dict(device_id=0, chunk_size=512, n_channels=2, fft_size=1024, history_chunks=2, lookback=640,
                    out_offset=256, ring_slots=0, sample_format=0)
        base.update(kw)
        cfg = _capi.AdspConfig(*[base[f[0]] for f in _capi.AdspConfig._fields_])
        h = ctypes.c_void_p()
        return lib.adsp_create(ctypes.byref(cfg), ctypes.byref(h)), h

    # lookback 704 / out_offset 192 are multiples of 2*threads_per_transform (64) but not of N/4 (128): the specialised
    # kernels resolve window and kept-slice phases in quarter chunks, so the ABI refuses them
    for bad in (dict(chunk_size=3), dict(chunk_size=502, sample_format=1), dict(fft_size=500), dict(fft_size=65536), dict(n_channels=0), dict(history_chunks=0), dict(lookback=641),
                dict(lookback=4096), dict(out_offset=1000), dict(out_offset=768, lookback=640), dict(ring_slots=2),
                dict(lookback=704), dict(out_offset=192),
                dict(device_id=99), dict(sample_format=7)):
        rc, h = create(**bad)
        assert rc == _capi.ADSP_ERR_ARG and not h.value, bad
        assert len(lib.adsp_last_error()) > 10
    rc, h = create()
    assert rc == 0 and h.value
    buf = (ctypes.c_float * (2 * 2 * 512))()  # room for two steps of [2 channels][512]
    assert lib.adsp_apply_host(h, buf, buf, 1) == _capi.ADSP_ERR_STATE  # no spectrum yet
    assert b"adsp_set_spectrum" in lib.adsp_last_error()
    spec = (ctypes.c_float * (2 * 513))()
    assert lib.adsp_set_spectrum(h, spec, 512) == _capi.ADSP_ERR_ARG  # wrong number of bins
    assert lib.adsp_set_spectrum(h, spec, 513) == 0
    assert lib.adsp_apply_host(h, buf, buf, 0) == _capi.ADSP_ERR_ARG
    assert lib.adsp_apply_host(h, None, buf, 1) == _capi.ADSP_ERR_ARG
    assert lib.adsp_set_block_outputs(h, 4096) == _capi.ADSP_ERR_ARG
    assert lib.adsp_set_block_outputs(h, 192) == _capi.ADSP_ERR_ARG    # not a multiple of N/4
    assert lib.adsp_set_block_outputs(h, 640) == 0 and lib.adsp_set_block_outputs(h, 512) == 0
    assert lib.adsp_set_kernel_reach(h, 1024) == _capi.ADSP_ERR_ARG and lib.adsp_set_kernel_reach(h, -1) == 0
    assert lib.adsp_apply_host(h, buf, buf, 1) == 0  # zero spectrum -> zero output, engine still healthy
    assert not any(buf)
    ms, n = ctypes.c_double(), ctypes.c_int()
    assert lib.adsp_enable_kernel_timing(h, 1) == 0 and lib.adsp_apply_host(h, buf, buf, 2) == 0
    assert lib.adsp_kernel_time(h, ctypes.byref(ms), ctypes.byref(n)) == 0 and n.value == 1 and ms.value > 0
    assert lib.adsp_destroy(h) == 0 and lib.adsp_destroy(None) == 0


@pytest.mark.parametrize("n,kind", [(256, "lowcut"), (1024, "eq"), (1000, "lowcut")])
def test_soak_mixed_call_types_ring_wrap(adsp, n, kind):
    """60 steps through one engine with every entry point interleaved (host single/multi-step, device single/multi-step,
    zero-copy ring): the history ring wraps many times, side-stream copies overlap kernels; one oracle stream."""
    import ctypes
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    o = orc()
    fs, channels, steps = 44100, 3, 60
    taps = design.lowcut_kernel(500, fs, n) if kind == "lowcut" else design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n)
    eng = FirEngine(FirStream(taps, n), channels=channels, ring_slots=5)
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, (steps, channels, n)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros_like(xd)
    s = torch.cuda.current_stream().cuda_stream
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    k, pattern = 0, ["host1", "dev3", "ring", "dev1", "host4", "ring", "ring", "dev5", "host1", "dev2"]
    i = 0
    while k < steps:
        op = pattern[i % len(pattern)]
        i += 1
        cnt = min(int(op[-1]) if op[-1].isdigit() else 1, steps - k)
        if op.startswith("host"):
            torch.cuda.synchronize()
            yd[k:k + cnt] = torch.from_numpy(eng.apply_host(x[k:k + cnt])).cuda()
        elif op.startswith("dev"):
            eng.apply_device(xd[k:k + cnt], yd[k:k + cnt], cnt, s)
        else:
            slot = eng.ring_acquire()
            assert hip.hipMemcpyAsync(slot, xd[k].data_ptr(), channels * n * 4, 3, s) == 0
            eng.apply_ring(yd[k], s)
        k += cnt
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    for c in range(channels):
        assert_parity(y[:, c].reshape(-1), o.direct_stream_convolution(taps, x[:, c].reshape(-1), n), what=f"N={n} ch {c}")


def test_plain_c_program_through_the_abi(adsp, tmp_path):
    """examples/capi_demo.c - no Python between the caller and libadsp: the reference's LowCut(800) design restated in C,
    3 channels x 8 chunks through adsp_apply_host, checked in C against the float64 direct convolution."""
    import os
    import subprocess
    from conftest import ROOT
    pkg = os.path.join(ROOT, "pyaudiodsptools_amd")
    exe = str(tmp_path / "capi_demo")
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "capi_demo.c"),
           "-L" + pkg, "-ladsp", "-lm", "-Wl,-rpath," + pkg, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "OK" in run.stdout and "real-spectrum stage 1" in run.stdout, run.stdout + run.stderr


@pytest.mark.parametrize("n,kind", [(512, "eq"), (4096, "lowcut")])
def test_two_stream_ring_pattern_with_a_real_producer(adsp, n, kind):
    """include/adsp.h: consecutive ring steps on two HIP streams in turn, each step's producer (a device copy into the
    ring slot) on the step's own stream, default ring length (2 x history = history + 2 slots): no extra synchronisation,
    every channel equals the exact engine over 40 steps (the ring wraps ten times)."""
    import ctypes
    import torch
    from pyaudiodsptools_amd import FirEngine, FirStream, design
    fs, channels, steps = 44100, 96, 40
    taps = design.lowcut_kernel(500, fs, n) if kind == "lowcut" else design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n)
    fir = FirStream(taps, n)
    eng = FirEngine(fir, channels=channels)  # ring_slots = 0 -> 2 x history
    assert eng.ring_slots >= eng.geometry.history_chunks + 2
    x = torch.empty((steps, channels, n), device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(n))
    y = torch.full_like(x, float("nan"))
    torch.cuda.synchronize()   # the fill runs on torch's default stream: finished before another stream writes into the buffer
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    for k in range(steps):
        s = streams[k % 2].cuda_stream
        slot = eng.ring_acquire()
        assert hip.hipMemcpyAsync(slot, x[k].data_ptr(), channels * n * 4, 3, s) == 0  # the producer, on the step's stream
        eng.apply_ring(y[k], s)
    torch.cuda.synchronize()
    ex = adsp.ExactFirEngine(fir, channels=channels)
    truth = torch.empty_like(x)
    ex.apply_device(x, truth, steps, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    assert float((y - truth).abs().max()) <= 1e-5 * float(truth.abs().max())


@pytest.mark.parametrize("n,m,lookahead,expect_real", [
    (512, 255, 127, True),      # the reference's shape: delay + centre = N
    (512, 129, 64, True),       # shorter symmetric kernel, delay + centre = 512
    (512, 129, 60, False),      # symmetric, but delay + centre is no multiple of N/4: falls back to the complex stage
    (1024, 511, 255, True),
    (4096, 1001, 500, True),
    (4096, 2047, 1023, True),
    (1000, 499, 249, True),     # generic geometry: delay + centre = 1000, a multiple of 4
    (1000, 499, 247, False),    # generic, delay + centre = 1002
    (3000, 1499, 749, True),
    (512, 254, 127, False),     # even length: no centre tap
])
def test_symmetric_kernels_take_the_real_spectrum_stage_and_match_direct_convolution(adsp, n, m, lookahead, expect_real):
    """Arbitrary symmetric FIRs (not only the reference's windowed sincs): geometry choice and parity."""
    from pyaudiodsptools_amd import FirStream
    rng = np.random.default_rng(n + m)
    half = rng.standard_normal((m + 1) // 2)
    taps = np.concatenate([half, half[::-1][m % 2:]]) / m
    assert len(taps) == m and np.array_equal(taps, taps[::-1])
    fir = FirStream(taps, n, latency_chunks=1, lookahead=lookahead)
    channels, steps = 3, 7
    eng = adsp.FirEngine(fir, channels=channels, optimize_for="batch")
    assert eng.geometry.zero_phase == expect_real and eng.real_spectrum == expect_real
    x = seeded_stream(900 + n + m, steps * channels * n).reshape(steps, channels, n)
    y = np.concatenate([eng.apply_host(x[:3]), eng.apply_host(x[3:4]), eng.apply_host(x[4:])])
    for c in range(channels):
        want = orc().direct_stream_convolution(taps, x[:, c].reshape(-1), n, 1, lookahead)
        assert_parity(y[:, c].reshape(-1), want, what=f"N={n} m={m} ch {c}")
    # an asymmetric kernel of the same length never takes it
    skew = taps.copy()
    skew[0] += 0.01
    eng2 = adsp.FirEngine(FirStream(skew, n, 1, lookahead), channels=1)
    assert not eng2.geometry.zero_phase and not eng2.real_spectrum
