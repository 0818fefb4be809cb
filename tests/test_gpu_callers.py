"""GPU parity of the callers that embed the FFT filters: CreateDelay and the reverb's delay lines, against
reference goldens (tests/golden/kat_callers.npz) and the CPU oracle.  Run with -m gpu on MI355X."""
import numpy as np
import pytest

from conftest import assert_parity, load_golden, seeded_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def adsp():
    import pyaudiodsptools_amd as pkg
    from pyaudiodsptools_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback by design)"
    return pkg


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat_callers")


DELAY = {
    # golden name: (fs, chunk, seed, chunks, kwargs, bit-exact?)  - exact when taps are at least a chunk apart
    "delay4096_default": (44100, 4096, 130, 14, {}, True),
    "delay4096_wet": (44100, 4096, 130, 14, {"wet": True}, True),
    "delay4096_300ms_4loops": (44100, 4096, 130, 14, {"time_in_ms": 300, "feedback_loops": 4}, True),
    "delay512_10ms_5loops": (44100, 512, 131, 12, {"time_in_ms": 10, "feedback_loops": 5}, False),
    "delay512_7ms_wet": (44100, 512, 131, 12, {"time_in_ms": 7.3, "feedback_loops": 1, "wet": True}, True),
    "delay512_noloops": (44100, 512, 131, 3, {"time_in_ms": 100, "feedback_loops": 0}, True),
}


@pytest.mark.parametrize("name", sorted(DELAY))
def test_delay_dropin_matches_reference_golden(adsp, kat, name):
    fs, n, seed, chunks, kw, exact = DELAY[name]
    adsp.config.initialize(fs, n)
    d = adsp.CreateDelay(**kw)
    x = seeded_stream(seed, chunks * n)
    keep = x.copy()
    outs = [d.apply(x[i * n:(i + 1) * n]) for i in range(chunks)]
    assert np.array_equal(x, keep), "the caller's chunks are not modified (the reference adds into them in place)"
    got = np.concatenate(outs)
    assert got.dtype == np.float32
    if exact:
        assert np.array_equal(got, kat[name]), name
    else:
        assert_parity(got, kat[name], what=name)
    # reset() returns to the zero history; one multi-step call equals the chunk-by-chunk stream
    d.reset()
    again = d.apply_batch(x.reshape(chunks, 1, n)).reshape(-1)
    assert np.array_equal(again, got)


def test_delay_attributes_follow_the_reference(adsp):
    adsp.config.initialize(44100, 4096)
    d = adsp.CreateDelay()
    assert d.time_in_samples == 22050 and d.max_samples == 88200 and d.wet is False
    assert np.array_equal(d.feedback_ramp, np.linspace(0.5, 0.1, 2, dtype=np.float32))
    assert d.line.history_chunks == 11  # ceil(44100 / 4096)
    with pytest.raises(ValueError):
        d.apply(np.zeros(100, np.float32))


def test_delay_many_channels_device_batches_and_straddling_taps(adsp):
    """[steps, C, N] device batches split unevenly; odd delays make 16-byte gathers straddle chunk boundaries."""
    import torch
    from oracle import callers_oracle as co
    n, C, steps = 64, 5, 23
    delays, gains = [1, 3, 62, 63, 64, 65, 127, 130, 1000], [0.9, -0.8, 0.7, 0.6, -0.5, 0.4, 0.3, 0.2, 0.1]
    line = adsp.DelayLine(delays, gains, dry=0.25, chunk_size=n, channels=C)
    x = seeded_stream(150, steps * C * n).reshape(steps, C, n)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    at = 0
    for k in (1, 2, 7, 1, 12):
        line.apply_device(d_in[at:at + k], d_out[at:at + k], k)
        at += k
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for c in range(C):
        xs = x[:, c].reshape(-1).astype(np.float64)
        want = 0.25 * xs
        for d, g in zip(delays, gains):
            want[d:] += np.float32(g).astype(np.float64) * xs[:-d]
        assert_parity(got[:, c].reshape(-1), want, what=f"channel {c}")
    with pytest.raises(RuntimeError):
        line.apply_device(d_in, d_in, steps)  # in place is refused
    with pytest.raises(RuntimeError):
        adsp.DelayLine([0], [1.0], chunk_size=n)
    with pytest.raises(RuntimeError):
        adsp.DelayLine([5], [1.0], chunk_size=30)


def test_delay_with_filters_enabled_runs_filter_then_taps(adsp):
    """The reference raises AttributeError here (EffectDelay.py:56,58); the defined behaviour is its reverb line's."""
    import torch
    from oracle import callers_oracle as co
    n, chunks = 512, 10
    adsp.config.initialize(44100, n)
    d = adsp.CreateDelay(10, 3, 200, 8000, True, True)
    o = co.OracleDelay(44100, n, 10, 3, 200, 8000, True, True)
    x = seeded_stream(151, chunks * n)
    got = np.concatenate([d.apply(x[i * n:(i + 1) * n]) for i in range(chunks)])
    want = np.concatenate([o.apply(x[i * n:(i + 1) * n]) for i in range(chunks)])
    assert_parity(got, want, what="host chain")
    d.reset()
    d_in = torch.from_numpy(x.reshape(chunks, 1, n)).cuda()
    d_out = torch.empty_like(d_in)
    d.apply_device(d_in, d_out, chunks)
    torch.cuda.synchronize()
    assert_parity(d_out.cpu().numpy().reshape(-1), want, what="device chain")


@pytest.mark.parametrize("name,fs,n,seed,chunks,ms", [("reverb512_default", 44100, 512, 132, 40, 1500),
                                                      ("reverb256_800ms_48k", 48000, 256, 133, 60, 800)])
def test_reverb_matches_reference_golden(adsp, kat, name, fs, n, seed, chunks, ms):
    from pyaudiodsptools_amd.delay import CreateReverb
    adsp.config.initialize(fs, n)
    rv = CreateReverb(ms)
    x = seeded_stream(seed, chunks * n)
    got = np.concatenate([rv.applyreverb(x[i * n:(i + 1) * n]) for i in range(chunks)])
    assert_parity(got, kat[name], what=name)
    rv.reset()
    again = rv.apply_batch(x.reshape(chunks, 1, n)).reshape(-1)
    assert_parity(again, kat[name], what=name + " one launch")


def test_reverb_any_chunk_size_many_channels(adsp):
    """Chunk 4096 overflows the reference's buffers; here it is just another shape.  Checked against the oracle's
    tap tables applied to oracle-filtered streams."""
    from oracle import callers_oracle as co
    from oracle import fftfilter_oracle as orc
    from pyaudiodsptools_amd.delay import CreateReverb
    fs, n, C, steps = 44100, 4096, 2, 6
    adsp.config.initialize(fs, n)
    rv = CreateReverb(1500, channels=C)
    x = seeded_stream(152, steps * C * n).reshape(steps, C, n)
    got = rv.apply_batch(x)
    total = int(1.5 * fs)
    for c in range(C):
        want = np.zeros(steps * n)
        for loops, cutoff in ((100, 5000), (50, 150)):
            hc = orc.OracleHighCut(cutoff, fs, n)
            y = np.concatenate([hc.apply(x[s, c]) for s in range(steps)]).astype(np.float64)
            for d, g in co.tap_table(total // loops, np.linspace(0.3, 0.01, loops, dtype=np.float32)[:loops - 1]):
                if d < len(y):
                    want[d:] += g * y[:-d]
        assert_parity(got[:, c].reshape(-1), want, what=f"channel {c}")
