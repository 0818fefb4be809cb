"""Round 6: reference CALL PATTERNS that had no fixture - Example4's in-place loop (history aliased by the caller's
array), Example3's callback thread, non-finite samples - and the round's engine work.  Run with -m gpu on MI355X."""
import threading

import numpy as np
import pytest

from conftest import ROOT, assert_parity, load_golden, seeded_stream  # noqa: F401

pytestmark = pytest.mark.gpu

SEEDS = {"LC": 201, "HC": 202, "EQ": 203}


@pytest.fixture(scope="module")
def adsp():
    import pyaudiodsptools_amd as pkg
    from pyaudiodsptools_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback by design)"
    return pkg


def orc():
    from oracle import fftfilter_oracle as o
    return o


def _make(adsp, tag, **kw):
    return {"LC": lambda: adsp.CreateLowCutFilterGPU(300, **kw), "HC": lambda: adsp.CreateHighCutFilterGPU(8000, **kw),
            "EQ": lambda: adsp.CreateEQ3BandFFTGPU(100, 2, 700, -4, 8000, 5, **kw)}[tag]()


# ---------------------------------------------------------------------------------------------------------------------
# 1. Example4's loop: split_data = array(MakeChunks(x)); split_data[i] = device.apply(split_data[i])   (Example4.py:9,18-19)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["LC", "HC", "EQ"])
@pytest.mark.parametrize("n,chunks,dec", [(512, 8, 1), (88200, 4, 64)])
@pytest.mark.parametrize("carrier", ["numpy", "torch"])
def test_example4_in_place_loop(adsp, tag, n, chunks, dec, carrier):
    """The reference's devices keep views of the rows as history, so its result is a filter over its own previous OUTPUTS
    (kat_inplace.npz, from the real reference).  Default here: the caller's array is copied, the loop returns the clean FIR stream -
    asserted against the reference's clean stream, with the measured distance to the reference's in-place result on record;
    alias_history=True: the reference's in-place result."""
    import torch
    kat = load_golden("kat_inplace")
    want_inplace, want_clean = kat[f"{tag}{n}_inplace"], kat[f"{tag}{n}_clean"]
    adsp.config.initialize(44100, n)
    x = seeded_stream(SEEDS[tag] + n, chunks * n)

    def loop(dev):
        if carrier == "numpy":
            arr = np.array(orc().make_chunks(x.copy(), n))
            for i in range(len(arr)):
                arr[i] = dev.apply(arr[i])
            return arr.reshape(-1)[::dec]
        arr = torch.from_numpy(x.copy()).cuda().reshape(chunks, n)   # cupy.array(MakeChunks(...)) in the reference
        for i in range(len(arr)):
            arr[i] = dev.apply(arr[i])
        return arr.cpu().numpy().reshape(-1)[::dec]

    try:
        clean = loop(_make(adsp, tag))
        assert_parity(clean, want_clean, what=f"{tag}{n} default (history copied): the FIR stream")
        apart = np.abs(clean - want_inplace).max() / np.abs(want_clean).max()
        assert apart > 0.5, "the documented divergence: the reference's in-place loop is full-scale away from the FIR stream"
        aliased = loop(_make(adsp, tag, alias_history=True))
        # The loop is a FEEDBACK system (every call re-filters the previous calls' outputs), so a rounding difference of one call is
        # filtered again by each later one: the whole loop is held to north_star's bound, max|d| <= 1e-5 x the magnitude of the data in
        # the transform windows (the loop annihilates most of the signal - LowCut at N = 88200 leaves 0.026 of a full-scale input -
        # while every window still holds full-scale input samples, whose float32 rounding is what the error is made of) ...
        bound = 1e-5 * max(np.abs(want_inplace).max(), np.abs(x).max())
        assert np.abs(aliased - want_inplace).max() <= bound, f"{tag}{n} alias_history=True: the in-place loop"
        # ... and every single call, given the reference's own (out[k-2], out[k-1], x[k]), to the suite's full criterion
        if dec == 1:
            ref_rows, xs = want_inplace.reshape(chunks, n), x.reshape(chunks, n)
            zero = np.zeros(n, np.float32)
            for k in range(chunks):
                dev = _make(adsp, tag, alias_history=True)
                dev.apply(ref_rows[k - 2].copy() if k >= 2 else zero)
                dev.apply(ref_rows[k - 1].copy() if k >= 1 else zero)
                assert_parity(dev.apply(xs[k]), ref_rows[k], what=f"{tag}{n} call {k} on the reference's own history")
    finally:
        adsp.config.initialize(44100, 4096)


def test_alias_history_follows_any_later_write_to_a_passed_chunk(adsp):
    """Not only Example4's pattern: whatever the caller writes into an array it has passed is what the next two calls transform
    (EffectFFTFilter.py:63-68) - checked against the oracle, which keeps references like the reference does."""
    n = 512
    adsp.config.initialize(44100, n)
    rng = np.random.default_rng(5)
    dev, ref = adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5, alias_history=True), orc().OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, 44100, n)
    bufs = [rng.uniform(-1, 1, n).astype(np.float32) for _ in range(6)]
    twin = [b.copy() for b in bufs]
    for k in range(6):
        got, want = dev.apply(bufs[k]), ref.apply(twin[k])
        assert_parity(got, want, what=f"call {k}")
        bufs[k] *= np.float32(0.5)      # scale the chunk just passed ...
        twin[k] *= np.float32(0.5)
        if k >= 1:
            bufs[k - 1][::7] = 0.25     # ... and poke into the one before
            twin[k - 1][::7] = 0.25
    with pytest.raises(ValueError):
        adsp.CreateLowCutFilter(300, channels=4, alias_history=True)
    adsp.config.initialize(44100, 4096)


# ---------------------------------------------------------------------------------------------------------------------
# 2. Non-finite samples
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["LC", "HC", "EQ"])
@pytest.mark.parametrize("vname,value", [("nan", np.nan), ("pinf", np.inf), ("ninf", -np.inf)])
@pytest.mark.parametrize("carrier", ["numpy", "torch", "mixed"])
def test_apply_poisons_the_calls_the_reference_poisons(adsp, tag, vname, value, carrier):
    """One NaN / Inf sample: the reference returns all-NaN chunks from the call that takes it and the two after it, and clean chunks
    again from the third (kat_nonfinite.npz).  apply() does the same for host chunks, device-resident chunks and a mixture."""
    import torch
    nf = load_golden("kat_nonfinite")
    n, chunks, where = 512, 8, int(nf["position"][0])
    adsp.config.initialize(44100, n)
    x = seeded_stream(SEEDS[tag] + 7, chunks * n)
    x[where] = value
    dev = _make(adsp, tag)
    outs = []
    for k in range(chunks):
        chunk = x[k * n:(k + 1) * n]
        on_gpu = carrier == "torch" or (carrier == "mixed" and k in (2, 4, 5))
        y = dev.apply(torch.from_numpy(chunk).cuda() if on_gpu else chunk)
        outs.append(y.cpu().numpy() if on_gpu else y)
    out = np.stack(outs)
    assert np.array_equal((~np.isfinite(out)).sum(axis=1), nf[f"{tag}_{vname}_nonfinite_per_call"])
    assert np.array_equal(np.isnan(out).sum(axis=1), nf[f"{tag}_{vname}_nan_per_call"])
    assert_parity(out[np.isfinite(out).all(axis=1)].reshape(-1), nf[f"{tag}_{vname}_finite_calls"], what="the finite calls")
    adsp.config.initialize(44100, 4096)


@pytest.mark.parametrize("n,channels", [(512, 5), (4096, 3), (88200, 2)])
def test_batches_poison_between_the_fir_support_and_the_references_three_chunks(adsp, n, channels):
    """apply_batch (not a reference entry point) lets the kernels' arithmetic decide: the non-finite outputs are a SUPERSET of the
    samples the FIR's support reaches from the bad sample and a SUBSET of the reference's three chunks; other channels stay clean."""
    adsp.config.initialize(44100, n)
    chunks = 7
    x = seeded_stream(n + channels, chunks * channels * n).reshape(chunks, channels, n).copy()
    pos = 2 * n + (3 * n) // 5          # absolute sample index of the bad sample in channel 1
    x[pos // n, 1, pos % n] = np.nan
    dev = adsp.CreateLowCutFilter(300, channels=channels)
    with np.errstate(all="ignore"):
        y = np.stack([dev.apply_batch(x[k]) for k in range(chunks)])
    bad = ~np.isfinite(y)
    assert not bad[:, [c for c in range(channels) if c != 1]].any(), "channels are independent"
    bad1 = bad[:, 1].reshape(-1)
    taps, d = dev.filter_length, dev.filter_length // 2
    # out[tau] = sum_t c[t] s[tau - N + d - t]: the sample at `pos` reaches tau in [pos + N - d, pos + N - d + taps)
    lo, hi = pos + n - d, pos + n - d + taps
    assert bad1[lo:hi].all(), "every output the FIR's support reaches is non-finite"
    k = pos // n
    assert not bad1[:k * n].any() and not bad1[(k + 3) * n:].any(), "nothing outside the reference's three chunks"
    adsp.config.initialize(44100, 4096)


# ---------------------------------------------------------------------------------------------------------------------
# 3. Example3's call pattern: apply() from a thread that did not create the device (PortAudio's callback thread, Example3.py:20-24)
# ---------------------------------------------------------------------------------------------------------------------
def test_apply_from_a_thread_that_did_not_create_the_device(adsp):
    n = 512
    adsp.config.initialize(44100, n)
    dev = adsp.CreateLowCutFilter(200)                       # created on the main thread (Example3.py:13)
    kat = load_golden("kat_streams")["D"]                    # LowCut(200), seed 1234, 6 chunks, from the reference
    x = seeded_stream(1234, 6 * n)
    long_x = seeded_stream(4242, 200 * n)
    ref = orc().OracleLowCut(200, 44100, n)
    want_long = np.concatenate([ref.apply(long_x[i * n:(i + 1) * n]) for i in range(200)])
    result = {}

    def callback_thread():
        try:
            result["D"] = np.concatenate([dev.apply(x[i * n:(i + 1) * n]) for i in range(6)])
            dev.reset()
            result["long"] = np.concatenate([dev.apply(np.frombuffer(long_x[i * n:(i + 1) * n].tobytes(), dtype=np.float32)) for i in range(200)])
        except BaseException as exc:  # noqa: BLE001 - reported on the main thread
            result["error"] = exc

    t = threading.Thread(target=callback_thread)
    t.start()
    t.join(120)
    assert not t.is_alive() and "error" not in result, result.get("error")
    assert_parity(result["D"], kat, what="golden D from a second thread")
    assert_parity(result["long"], want_long, what="200 calls from a second thread")
    # ... and the main thread continues the same stream afterwards (state lives in the engine, not in the thread)
    more = seeded_stream(4243, 2 * n)
    got = np.concatenate([dev.apply(more[:n]), dev.apply(more[n:])])
    want = np.concatenate([ref.apply(more[:n]), ref.apply(more[n:])])
    assert_parity(got, want, what="main thread continues")
    adsp.config.initialize(44100, 4096)


# ---------------------------------------------------------------------------------------------------------------------
# 4. The long-kernel engine's surface: checkpoint / resume, synchronise, calls on changing streams, the tail-only ring
# ---------------------------------------------------------------------------------------------------------------------
def _long_fir(adsp, n=88200, taps_len=44099, seed=3, latency=1):
    rng = np.random.default_rng(seed)
    taps = rng.standard_normal(taps_len) * np.hanning(taps_len) / np.sqrt(taps_len) * 3.0
    return adsp.FirStream(taps, n, latency_chunks=latency, lookahead=taps_len // 2), taps


@pytest.mark.parametrize("carry", [-1, 1])
@pytest.mark.parametrize("block", [8192, 16384])
def test_upols_state_roundtrip_reset_and_refusals(adsp, block, carry):
    """Mirrors test_multistep_equals_streaming_and_state_roundtrip for the partitioned engine: a checkpoint taken mid-stream and
    restored into a FRESH engine continues the stream bit for bit; reset returns to the zero state; a state of another shape is refused."""
    fir, _ = _long_fir(adsp)
    n, channels, steps = fir.chunk_size, 3, 6
    x = seeded_stream(77, steps * channels * n).reshape(steps, channels, n)
    a = adsp.UpolsFirEngine(fir, channels=channels, block=block)
    a.set_carry(carry)
    whole = np.concatenate([a.apply_host(x[k:k + 1]) for k in range(steps)])
    b = adsp.UpolsFirEngine(fir, channels=channels, block=block)
    b.set_carry(carry)
    part1 = np.concatenate([b.apply_host(x[k:k + 1]) for k in range(3)])
    state = b.get_state()
    assert state.dtype == np.uint8 and state.size > b.delay_line_bytes
    part2 = np.concatenate([b.apply_host(x[k:k + 1]) for k in range(3, steps)])
    assert np.array_equal(np.concatenate([part1, part2]), whole)
    c = adsp.UpolsFirEngine(fir, channels=channels, block=block)
    c.set_carry(carry)
    c.set_state(state)   # (what b carried over its third call's end is not part of the state: c computes that block itself - the same samples)
    assert np.array_equal(np.concatenate([c.apply_host(x[k:k + 1]) for k in range(3, steps)]), part2), "resume is bit for bit"
    c.reset()
    assert np.array_equal(np.concatenate([c.apply_host(x[k:k + 1]) for k in range(3)]), part1), "after reset"
    other = adsp.UpolsFirEngine(fir, channels=channels + 1, block=block)
    with pytest.raises(adsp.AdspError):
        other.set_state(state)
    with pytest.raises(adsp.AdspError):
        c.set_state(state[:1000])
    bad = state.copy()
    bad[0] ^= 0xFF
    with pytest.raises(adsp.AdspError):
        c.set_state(bad)
    old_build = state.copy()  # a checkpoint of the builds whose delay line kept UNSPLIT spectra (magic "UPOL"): refused by name
    old_build[:4] = np.frombuffer(np.uint32(0x55504F4C).tobytes(), np.uint8)
    with pytest.raises(adsp.AdspError, match="unsplit"):
        c.set_state(old_build)
    for e in (a, b, c, other):
        e.close()


@pytest.mark.parametrize("fmt,effect", [("f32", None), ("f32", "tremolo"), ("f32", "saturator"), ("f32", "softclip"), ("s16", None)])
@pytest.mark.parametrize("block", [8192, 16384])
def test_upols_carry_equals_recompute(adsp, block, fmt, effect):
    """adsp_upols_set_carry: the block that straddles the end of a call computed once and carried (float32, before the effect) to the next
    call's output, against the same block computed in both calls: the SAME samples, bit for bit - float32 and int16, with a fused
    stateless effect and with the tremolo (whose table index the carried samples take from the call they are delivered in), over calls of
    1 - 3 chunks, a filter change (which drops what was carried) and a reset."""
    n, channels = 20000, 5
    fir, _ = _long_fir(adsp, n=n, taps_len=45001, seed=21, latency=3)
    fir2, _ = _long_fir(adsp, n=n, taps_len=45001, seed=22, latency=3)
    adsp.config.initialize(44100, n)
    calls = [1, 2, 1, 3, 1, 1, 2]
    total = sum(calls)
    if fmt == "s16":
        x = np.random.default_rng(5).integers(-9000, 9000, (total, channels, n)).astype(np.int16)
    else:
        x = seeded_stream(91, total * channels * n).reshape(total, channels, n)
    fx = {None: None, "tremolo": lambda: adsp.CreateTremolo(0.7, 3.3), "saturator": lambda: adsp.CreateSaturator(),
          "softclip": lambda: adsp.CreateSoftClipper()}[effect]
    outs = []
    for mode in (0, 1):
        eng = adsp.UpolsFirEngine(fir, channels=channels, sample_format=fmt, max_steps=3, block=block)
        eng.set_carry(mode)
        if fx is not None:
            eng.set_epilogue(fx())
        got, k = [], 0
        for i, m in enumerate(calls):
            if i == 4:
                eng.set_spectra(adsp.design.partition_uniform(fir2, block, eng.gain).spectra)   # the carried block belonged to the old filter
            got.append(eng.apply_host(x[k:k + m]))
            k += m
        eng.reset()
        got.append(eng.apply_host(x[:2]))
        outs.append(np.concatenate(got))
        eng.close()
    adsp.config.initialize(44100, 4096)
    assert outs[0].dtype == (np.int16 if fmt == "s16" else np.float32) and np.abs(outs[0].astype(np.float64)).max() > 0
    assert np.array_equal(outs[0], outs[1]), f"{int((outs[0] != outs[1]).sum())} samples differ"


def test_upols_calls_on_changing_streams_and_synchronize(adsp):
    """numpy chunks (NULL stream) and device tensors under non-default, non-blocking torch streams alternate on one
    long-kernel device; the library orders the calls itself (event behind every launch pair) and synchronize(stream) waits for the
    engine's last call wherever it went."""
    import torch
    fir, taps = _long_fir(adsp, n=20000, taps_len=30001, seed=9, latency=2)
    n, channels, steps = fir.chunk_size, 2, 8
    x = seeded_stream(78, steps * channels * n).reshape(steps, channels, n)
    ref = adsp.UpolsFirEngine(fir, channels=channels)
    want = np.concatenate([ref.apply_host(x[k:k + 1]) for k in range(steps)])
    eng = adsp.UpolsFirEngine(fir, channels=channels)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    xd = torch.from_numpy(x).cuda()
    outs = [None] * steps
    torch.cuda.synchronize()
    for k in range(steps):
        if k % 3 == 0:
            outs[k] = torch.from_numpy(eng.apply_host(x[k:k + 1])).cuda()
        else:
            s = streams[k % 2]
            outs[k] = torch.empty((1, channels, n), device="cuda")
            eng.apply_device(xd[k:k + 1], outs[k], 1, s.cuda_stream)
    eng.synchronize(streams[0].cuda_stream)   # the last call (k = 7) went to streams[1]: still waited for
    got = torch.cat(outs).cpu().numpy()
    assert np.array_equal(got, want)
    for c in range(channels):
        truth = orc().direct_stream_convolution(taps, x[:, c].reshape(-1), n, fir.latency_chunks, fir.lookahead)
        assert_parity(got[:, c].reshape(-1), truth, what=f"channel {c}")
    ref.close()
    eng.close()


@pytest.mark.parametrize("n,taps_len,latency,calls", [(88200, 44099, 1, [1, 1, 1, 1]), (3000, 20001, 9, [1, 7, 2, 30, 1, 1]),
                                                      (16, 9001, 1306, [600, 1, 1500, 3])])
def test_upols_ring_keeps_only_the_tail_that_later_windows_reach(adsp, n, taps_len, latency, calls):
    """The ring update copies the last 2B samples of a call's input (one workgroup per channel inside the multiply launch) instead of
    whole chunks: chunk sizes above, near and far below the block size, calls of many chunks, against the float64 direct sum."""
    import torch
    fir, taps = _long_fir(adsp, n=n, taps_len=taps_len, seed=n, latency=latency)
    channels, total = 5, sum(calls)
    x = seeded_stream(79 + n, total * channels * n).reshape(total, channels, n)
    for block in (8192, 16384):
        if fir.delay < block:
            continue
        eng = adsp.UpolsFirEngine(fir, channels=channels, block=block, max_steps=max(1, min(64, max(calls))))
        got, k = [], 0
        for m in calls:
            got.append(eng.apply_host(x[k:k + m]))
            k += m
        got = np.concatenate(got)
        for c in (0, channels - 1):
            truth = orc().direct_stream_convolution(taps, x[:, c].reshape(-1), n, latency, fir.lookahead)
            assert_parity(got[:, c].reshape(-1), truth, what=f"N={n} block {block} channel {c}")
        eng.close()
    torch.cuda.synchronize()


def test_long_kernels_shard_through_the_banks_and_take_a_broadcast_filter(adsp):
    """VERDICT r5 "missing" 4: Example4's shape (chunk 88200) through dist.LocalFirBank / dist.ShardedFirBank - the banks build the
    uniformly partitioned engine (make_engine) and the one collective carries its partition spectra: adsp_upols_bcast_spectra (one
    process), adsp_upols_bcast_spectra_rank (the "abi" carrier) and torch's broadcast + adsp_upols_set_spectra, each at world size 1."""
    from pyaudiodsptools_amd.dist import LocalFirBank, ShardedFirBank
    from pyaudiodsptools_amd.engine import rccl_finalize
    fir, taps = _long_fir(adsp)
    other, _ = _long_fir(adsp, seed=4)
    n, channels, steps = fir.chunk_size, 3, 3
    x = seeded_stream(80, steps * channels * n).reshape(steps, channels, n)
    want = [orc().direct_stream_convolution(taps, x[:, c].reshape(-1), n, 1, fir.lookahead) for c in range(channels)]

    def check(y, what):
        for c in range(channels):
            assert_parity(y[:, c].reshape(-1), want[c], what=f"{what}, channel {c}")

    bank = LocalFirBank(fir, channels, devices=[0])
    assert type(bank.engines[0]).__name__ == "UpolsFirEngine"
    check(np.concatenate([bank.apply_host(x[k:k + 1]) for k in range(steps)]), "LocalFirBank")
    bank.close()
    for carrier in ("torch", "abi"):
        sb = ShardedFirBank(fir, channels, device=0, carrier=carrier)
        assert type(sb.engine).__name__ == "UpolsFirEngine" and (sb.lo, sb.hi) == (0, channels)
        assert np.array_equal(sb.spectrum, sb.engine.get_spectra())
        check(np.concatenate([sb.engine.apply_host(x[k:k + 1]) for k in range(steps)]), f"ShardedFirBank, carrier {carrier}")
        sb.engine.close()
    # the filter of a running engine: another kernel of the same partitioning, then back
    eng = adsp.make_engine(fir, channels=channels)
    mine = eng.get_spectra()
    theirs = adsp.make_engine(other, channels=1)
    eng.set_spectra(theirs.get_spectra())
    assert np.array_equal(eng.get_spectra(), theirs.get_spectra()) and not np.array_equal(mine, eng.get_spectra())
    eng.set_spectra(mine)
    eng.reset()
    check(np.concatenate([eng.apply_host(x[k:k + 1]) for k in range(steps)]), "after set_spectra round trip")
    with pytest.raises(ValueError):
        eng.set_spectra(mine[:-2])
    eng.close()
    theirs.close()
    rccl_finalize()


# ---------------------------------------------------------------------------------------------------------------------
# 5. Cross products of the widening rows (VERDICT r5 "missing" 5): tremolo on long kernels, int16 on chunk sizes that are not multiples
#    of 4, WavBank.process reusing its engine
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,n,taps_len,latency,lfo_hz", [
    ("upols", 20000, 40001, 2, 4.5),             # LFO period 9800 samples: several periods per chunk
    ("upols", 20000, 40001, 2, 44100 / 20000),   # period == chunk: the reference's buffer quirk (every chunk replays the same table segment)
    ("upols", 20000, 40001, 2, 0.7),             # period 63000 > chunk
    ("partitioned", 10002, 36001, 3, 4.5),       # chunk size not a multiple of 4: one engine pass per kernel slice + the row-wise tremolo pass
    ("partitioned", 10002, 36001, 3, 44100 / 10002),
])
def test_tremolo_behind_a_long_kernel(adsp, kind, n, taps_len, latency, lfo_hz):
    """CreateTremolo applied to the output of a device whose kernel is longer than one transform (EffectTremolo.py:27-47 after
    EffectFFTFilter.apply): every channel's LFO in step, table index following the stream across calls of several chunks, the buffer
    quirk included - against the oracle's tremolo over the float64 direct sum."""
    from oracle import effects_oracle as fx
    adsp.config.initialize(44100, n)
    fir, taps = _long_fir(adsp, n=n, taps_len=taps_len, seed=11, latency=latency)
    channels, calls = 3, [1, 2, 1, 3]
    total = sum(calls)
    x = seeded_stream(81 + n, total * channels * n).reshape(total, channels, n)
    eng = adsp.make_engine(fir, channels=channels, **({"max_steps": 2} if kind == "upols" else {}))
    assert type(eng).__name__ == ("UpolsFirEngine" if kind == "upols" else "PartitionedFirEngine")
    eng.set_epilogue(adsp.CreateTremolo(0.6, lfo_hz))
    got, k = [], 0
    for m in calls:
        got.append(eng.apply_host(x[k:k + m]))
        k += m
    got = np.concatenate(got)
    for c in range(channels):
        clean = orc().direct_stream_convolution(taps, x[:, c].reshape(-1), n, latency, fir.lookahead).astype(np.float32)
        trem = fx.OracleTremolo(44100, 0.6, lfo_hz)
        want = np.concatenate([trem.apply(clean[j * n:(j + 1) * n]) for j in range(total)])
        assert_parity(got[:, c].reshape(-1), want, what=f"{kind} N={n} lfo {lfo_hz:.3f} channel {c}")
    # reset restarts filter history AND LFO
    eng.reset()
    again = eng.apply_host(x[:1])
    assert np.array_equal(again, got[:1])
    eng.close()
    adsp.config.initialize(44100, 4096)


@pytest.mark.parametrize("n,taps_len,latency", [(1002, 500, 1), (30, 14, 1), (10002, 36001, 3)])
def test_int16_batches_on_chunk_sizes_that_are_not_multiples_of_four(adsp, n, taps_len, latency):
    """make_engine(sample_format="s16") where the int16 kernels' 8-byte accesses cannot tile the chunk (N % 4 != 0), single transform
    and long kernel: the float32 dword-access kernels behind two conversion passes, against the exact int16 engine (float64 direct sum
    with the reference's conversions to the letter) to within one LSB."""
    import torch
    fir, _ = _long_fir(adsp, n=n, taps_len=taps_len, seed=13, latency=latency)
    fir = adsp.FirStream(fir.taps / np.abs(fir.taps).sum() * 0.9, n, latency_chunks=latency, lookahead=fir.lookahead)  # |y| < 1: no int16 wrap
    channels, steps = 3, 5
    pcm = np.random.default_rng(n).integers(-30000, 30000, (steps, channels, n)).astype(np.int16)
    eng = adsp.make_engine(fir, channels=channels, sample_format="s16")
    assert type(eng).__name__ == "Pcm16Adapter"
    got = eng.apply_host(pcm)
    assert got.dtype == np.int16 and got.shape == pcm.shape
    ex = adsp.ExactFirEngine(fir, channels=channels, sample_format="s16")
    want = ex.apply_host(pcm)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02, (diff.max(), (diff > 0).mean())
    # the device path (torch int16 tensors) continues the same stream
    eng.reset()
    d_in, d_out = torch.from_numpy(pcm).cuda(), torch.empty((steps, channels, n), dtype=torch.int16, device="cuda")
    eng.apply_device(d_in, d_out, steps)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), got)
    with pytest.raises(TypeError):
        eng.apply_host(pcm.astype(np.float32))
    eng.close()
    ex.close()


def test_wavbank_process_reuses_its_engine(adsp, tmp_path):
    """WavBank.process keeps the engine of the last calls (keyed on filter, channel count, device, mode): the second call with the same
    filter creates nothing, starts from zero history like the first and returns the same samples; a long kernel (chunk 88200) goes through
    the same front end."""
    import wave
    from pyaudiodsptools_amd import wavio
    wavio.close_bank_engines()
    adsp.config.initialize(44100, 4096)
    rng = np.random.default_rng(21)
    paths = []
    for i in range(3):
        p = str(tmp_path / f"f{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1 + i % 2)
            w.setsampwidth(2)
            w.setframerate(44100)
            w.writeframes(rng.integers(-20000, 20000, (30000 + 1000 * i) * (1 + i % 2)).astype(np.int16).tobytes())
        paths.append(p)
    bank = adsp.WavBank(paths)
    fir = adsp.CreateLowCutFilter(800).fir
    created = []
    real = wavio.make_engine
    wavio.make_engine = lambda *a, **k: (created.append(1), real(*a, **k))[1]
    try:
        first = bank.process(fir)
        second = bank.process(fir)
        assert len(created) == 1 and all(np.array_equal(a, b) for a, b in zip(first, second))
        other = bank.process(adsp.CreateHighCutFilter(8000).fir)
        assert len(created) == 2 and not np.array_equal(other[0], first[0])
        exact = bank.process(fir, exact=True)
        d = np.abs(exact[0].astype(np.int32) - first[0].astype(np.int32))
        assert d.max() <= 1
    finally:
        wavio.make_engine = real
    # Example4's chunk size through the bank: 44 099 taps, the uniformly partitioned int16 engine
    adsp.config.initialize(44100, 88200)
    long_bank = adsp.WavBank(paths[:2], chunk_size=88200)
    long_fir = adsp.CreateLowCutFilter(300).fir
    got = long_bank.process(long_fir)
    want = long_bank.process(long_fir, exact=True)
    for a, b in zip(got, want):
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1
    assert any(type(e).__name__ == "UpolsFirEngine" for e in wavio._bank_cache.values())
    wavio.close_bank_engines()
    adsp.config.initialize(44100, 4096)


def test_package_works_without_torch_in_the_process():
    """VERDICT r5 code #11: _capi.load() imports torch only to share ITS HIP runtime and RCCL copy when torch is there; a process
    in which torch cannot be imported loads libadsp.so against the system's ROCm and runs the reference's API (numpy in, numpy out)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.modules['torch'] = None\n"          # `import torch` now raises ImportError
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT + '/tests'!r})\n"
        "import numpy as np\n"
        "import pyaudiodsptools_amd as adsp\n"
        "from oracle import fftfilter_oracle as orc\n"
        "adsp.config.initialize(44100, 512)\n"
        "dev, ref = adsp.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, 44100, 512)\n"
        "x = np.random.default_rng(1).uniform(-1, 1, 6 * 512).astype(np.float32)\n"
        "got = np.concatenate([dev.apply(x[i * 512:(i + 1) * 512]) for i in range(6)])\n"
        "want = np.concatenate([ref.apply(x[i * 512:(i + 1) * 512]) for i in range(6)])\n"
        "err = np.abs(got - want).max() / np.abs(want).max()\n"
        "assert 'torch' not in [m for m in sys.modules if sys.modules[m] is not None], 'torch got imported'\n"
        "print('rel_err', err)\n"
        "assert err <= 1e-5\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rel_err" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]
