"""GPU parity of the recursive devices (last item): IIR 3-band EQ and compressor as per-channel scans,
bit for bit against reference goldens (tests/golden/kat_recursive.npz) and the CPU oracle.  Run with -m gpu on MI355X."""
import numpy as np
import pytest

from conftest import load_golden, seeded_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def adsp():
    import pyaudiodsptools_amd as pkg
    from pyaudiodsptools_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback by design)"
    return pkg


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat_recursive")


@pytest.mark.parametrize("band", ["low", "mid", "high"])
def test_iir_band_dropin_is_bit_exact(adsp, kat, band):
    n = 1024
    adsp.config.initialize(44100, n)
    eq = adsp.CreateEQ3Band(100, 2, 700, -4, 8000, 5)
    x = seeded_stream(160, 6 * n)
    keep = x.copy()
    f = getattr(eq, f"apply{band}band")
    got = np.concatenate([f(x[i * n:(i + 1) * n]) for i in range(6)])
    assert got.dtype == np.float32 and np.array_equal(x, keep)
    assert np.array_equal(got, kat["iir_" + band])


def test_iir_cascade_attributes_and_one_pass(adsp, kat):
    n = 1024
    adsp.config.initialize(44100, n)
    eq = adsp.CreateEQ3Band(250, -6, 1500, 3, 6000, -2.5)
    co = np.array([getattr(eq, b + k) for b in ("LOW", "MID", "HIGH") for k in ("b0", "b1", "b2", "a0", "a1", "a2")])
    assert np.array_equal(co, kat["iir_coeffs"])
    x = seeded_stream(160, 6 * n)
    got = np.concatenate([eq.applyhighband(eq.applymidband(eq.applylowband(x[i * n:(i + 1) * n]))) for i in range(6)])
    assert np.array_equal(got, kat["iir_cascade"])
    # the three sections in one kernel, all six chunks in one call
    one = eq.apply_all_batch(x.reshape(6, 1, n)).reshape(-1)
    assert np.array_equal(one, kat["iir_cascade"])
    with pytest.raises(ValueError):
        eq.applylowband(np.zeros(100, np.float32))


def test_iir_many_channels_ragged_tiles_device_in_place(adsp):
    """70 channels (two workgroups, the second ragged), chunk 100 (ragged time tiles), uneven device calls, in place."""
    import torch
    from oracle import recursive_oracle as ro
    n, C, steps = 100, 70, 9
    adsp.config.initialize(44100, n)
    eq = adsp.CreateEQ3Band(120, 4, 900, -3, 5000, 2, channels=C)
    x = seeded_stream(170, steps * C * n).reshape(steps, C, n)
    d = torch.from_numpy(x).cuda()
    at = 0
    for k in (1, 3, 5):
        eq.cascade.apply_device(d[at:at + k], d[at:at + k], k)
        at += k
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    for c in (0, 1, 63, 64, 69):
        o = ro.OracleEQ3Band(120, 4, 900, -3, 5000, 2)
        want = np.stack([o.applyhighband(o.applymidband(o.applylowband(x[s, c]))) for s in range(steps)])
        assert np.array_equal(got[:, c], want), c


COMP = {"default": {}, "fast": {"threshold_in_db": -20, "ratio": 0.3, "attack_in_ms": 0.5, "release_in_ms": 2.0},
        "slow": {"threshold_in_db": -10, "ratio": 0.8, "attack_in_ms": 10.0, "release_in_ms": 100.0}}


@pytest.mark.parametrize("tag", sorted(COMP))
def test_compressor_dropin_is_bit_exact(adsp, kat, tag):
    n = 1024
    adsp.config.initialize(44100, n)
    cp = adsp.CreateCompressor(**COMP[tag])
    x = kat["comp_input"]
    got = np.concatenate([cp.apply(x[i * n:(i + 1) * n]) for i in range(12)])
    assert np.array_equal(got, kat["comp_" + tag]), int(np.argmax(got != kat["comp_" + tag]))
    cp.reset()
    again = cp.apply_batch(x.reshape(12, 1, n)).reshape(-1)
    assert np.array_equal(again, kat["comp_" + tag])


def test_compressor_many_channels_against_oracle(adsp):
    from oracle import recursive_oracle as ro
    n, C, steps = 256, 67, 10
    adsp.config.initialize(48000, n)
    cp = adsp.CreateCompressor(-18, 0.5, 1.0, 7.5, channels=C)
    rng = np.random.default_rng(171)
    env = np.repeat(rng.choice([0.03, 0.1, 0.2, 0.8], size=(steps * C * n) // 32), 32).astype(np.float32)
    x = (rng.uniform(-1, 1, steps * C * n).astype(np.float32) * env).reshape(steps, C, n)
    got = cp.apply_batch(x)
    for c in (0, 31, 63, 64, 66):
        o = ro.OracleCompressor(48000, -18, 0.5, 1.0, 7.5)
        want = np.stack([o.apply(x[s, c]) for s in range(steps)])
        assert np.array_equal(got[:, c], want), c
    with pytest.raises(ValueError):
        adsp.CreateCompressor(attack_in_ms=0.0)


GATE = {"default": {}, "fast": {"threshold_in_db": -12, "depth": 0.25, "attack": 0.5, "release": 3.0},
        "deep": {"threshold_in_db": -20, "depth": 0.01, "attack": 10.0, "release": 50.0}}


@pytest.mark.parametrize("tag", sorted(GATE))
def test_gate_dropin_is_bit_exact(adsp, tag):
    """EffectGate.py:42-126 (its apply DOES return the shaped copy, :126) against vectors captured from the reference."""
    kat = load_golden("kat_gate")
    n = 1024
    adsp.config.initialize(48000, n)  # ignored by the gate like in the reference: envelopes are built for 44100 Hz
    g = adsp.CreateGate(**GATE[tag])
    x = kat["gate_input"]
    keep = x.copy()
    got = np.concatenate([g.apply(x[i * n:(i + 1) * n]) for i in range(16)])
    assert got.dtype == np.float32 and np.array_equal(x, keep)       # fresh array, the argument is left alone
    assert np.array_equal(got, kat["gate_" + tag]), int(np.argmax(got != kat["gate_" + tag]))
    g.reset()
    again = g.apply_batch(x.reshape(16, 1, n)).reshape(-1)
    assert np.array_equal(again, kat["gate_" + tag])


def test_gate_many_channels_against_oracle(adsp):
    from oracle import recursive_oracle as ro
    n, C, steps = 256, 67, 10
    adsp.config.initialize(44100, n)
    g = adsp.CreateGate(-9, 0.2, 1.0, 7.5, channels=C)
    rng = np.random.default_rng(172)
    env = np.repeat(rng.choice([0.03, 0.2, 0.5, 1.0], size=(steps * C * n) // 32), 32).astype(np.float32)
    x = (rng.uniform(-1, 1, steps * C * n).astype(np.float32) * env).reshape(steps, C, n)
    got = g.apply_batch(x)
    for c in (0, 31, 63, 64, 66):
        o = ro.OracleGate(-9, 0.2, 1.0, 7.5)
        want = np.stack([o.apply(x[s, c]) for s in range(steps)])
        assert np.array_equal(got[:, c], want), c
    with pytest.raises(RuntimeError):
        adsp.CreateGate(depth=-0.5)   # a non-positive depth is refused by the C ABI


def test_scan_engine_argument_errors(adsp):
    from pyaudiodsptools_amd import ScanEngine
    with pytest.raises(RuntimeError):
        ScanEngine.biquad(np.zeros((5, 5)), 64)          # more sections than ADSP_SCAN_MAX_SECTIONS
    with pytest.raises(RuntimeError):
        ScanEngine.biquad(np.zeros((1, 5)), 0)           # chunk size
    eng = ScanEngine.biquad([[1.0, 0.0, 0.0, 0.0, 0.0]], 8, channels=2)
    x = np.arange(32, dtype=np.float32).reshape(2, 2, 8)
    y = eng.apply_host(x)                               # pure one-sample delay per channel, across chunks
    assert np.array_equal(y[0, 0], np.concatenate([[0], x[0, 0, :-1]])) and y[1, 0, 0] == x[0, 0, -1]
    with pytest.raises(ValueError):
        eng.apply_host(np.zeros((2, 3, 8), np.float32))


@pytest.mark.parametrize("kind", ["compressor", "gate"])
@pytest.mark.parametrize("n,C,steps", [(100, 5, 7), (1000, 3, 4), (64, 9, 12), (4096, 2, 2), (37, 8200, 2)])
def test_time_across_lanes_equals_lane_per_channel_and_oracle(adsp, kind, n, C, steps):
    """Round 6: up to 8192 channels the compressor / gate run with TIME across the lanes of a wave (one ballot of threshold bits, the
    state machine walked in wave-uniform code, all 64 gains applied at once); beyond that - and with ADSP_SCAN_LANE_PER_CHANNEL set - one
    lane walks one channel.  Both forms against each other bit for bit on every channel, and against the oracle on a few: ragged tiles
    (N = 100, 1000, 37), one-tile chunks, envelopes too long for LDS (the gate's 300 ms release), more channels than the limit."""
    import os
    from oracle import recursive_oracle as ro
    adsp.config.initialize(44100, n)
    rng = np.random.default_rng(n * 31 + C)
    env = np.repeat(rng.choice([0.02, 0.1, 0.3, 1.0], size=-(-steps * C * n // 16)), 16)[:steps * C * n].astype(np.float32)
    x = (rng.uniform(-1, 1, steps * C * n).astype(np.float32) * env).reshape(steps, C, n)
    args = (-18, 0.5, 1.0, 7.5) if kind == "compressor" else (-9, 0.2, 1.0, 300.0 if n == 1000 else 7.5)
    make = (lambda: adsp.CreateCompressor(*args, channels=C)) if kind == "compressor" else (lambda: adsp.CreateGate(*args, channels=C))
    keep = os.environ.pop("ADSP_SCAN_LANE_PER_CHANNEL", None)
    try:
        dev = make()
        first = dev.apply_batch(x[:steps // 2])          # state carried from one call to the next
        got = np.concatenate([first, dev.apply_batch(x[steps // 2:])])
        os.environ["ADSP_SCAN_LANE_PER_CHANNEL"] = "1"
        other = make().apply_batch(x)
    finally:
        os.environ.pop("ADSP_SCAN_LANE_PER_CHANNEL", None)
        if keep is not None:
            os.environ["ADSP_SCAN_LANE_PER_CHANNEL"] = keep
        adsp.config.initialize(44100, 4096)
    assert np.array_equal(got, other), int(np.argmax(got != other))
    for c in sorted({0, C // 2, C - 1}):
        o = ro.OracleCompressor(44100, *args) if kind == "compressor" else ro.OracleGate(*args)
        want = np.stack([o.apply(x[s, c]) for s in range(steps)])
        assert np.array_equal(got[:, c], want), c
