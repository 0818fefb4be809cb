"""Pin oracle/recursive_oracle.py (IIR 3-band EQ, compressor) bit-exactly to vectors captured from the reference."""
import numpy as np
import pytest

from conftest import load_golden, seeded_stream
from oracle import recursive_oracle as ro


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat_recursive")


def test_rbj_coefficients_match_reference(kat):
    co = ro.rbj_coefficients(250, -6, 1500, 3, 6000, -2.5)
    got = np.array(co["low"] + co["mid"] + co["high"])
    assert np.array_equal(got, kat["iir_coeffs"])


@pytest.mark.parametrize("band", ["low", "mid", "high"])
def test_iir_band_is_bit_exact(kat, band):
    n = 1024
    x = seeded_stream(160, 6 * n)
    eq = ro.OracleEQ3Band(100, 2, 700, -4, 8000, 5)
    f = getattr(eq, f"apply{band}band")
    got = np.concatenate([f(x[i * n:(i + 1) * n]) for i in range(6)])
    assert got.dtype == np.float32 and np.array_equal(got, kat["iir_" + band])


def test_iir_cascade_is_bit_exact(kat):
    n = 1024
    x = seeded_stream(160, 6 * n)
    eq = ro.OracleEQ3Band(250, -6, 1500, 3, 6000, -2.5)
    got = np.concatenate([eq.applyhighband(eq.applymidband(eq.applylowband(x[i * n:(i + 1) * n]))) for i in range(6)])
    assert np.array_equal(got, kat["iir_cascade"])


COMP = {"default": {}, "fast": {"threshold_in_db": -20, "ratio": 0.3, "attack_in_ms": 0.5, "release_in_ms": 2.0},
        "slow": {"threshold_in_db": -10, "ratio": 0.8, "attack_in_ms": 10.0, "release_in_ms": 100.0}}


@pytest.mark.parametrize("tag", sorted(COMP))
def test_compressor_is_bit_exact(kat, tag):
    n = 1024
    x = kat["comp_input"]
    cp = ro.OracleCompressor(44100, **COMP[tag])
    got = np.concatenate([cp.apply(x[i * n:(i + 1) * n]) for i in range(12)])
    want = kat["comp_" + tag]
    assert np.array_equal(got, want), int(np.argmax(got != want))
    assert (want != x).mean() > 0.2  # the compressor actually worked on this input


GATE = {"default": {}, "fast": {"threshold_in_db": -12, "depth": 0.25, "attack": 0.5, "release": 3.0},
        "deep": {"threshold_in_db": -20, "depth": 0.01, "attack": 10.0, "release": 50.0}}


@pytest.mark.parametrize("tag", sorted(GATE))
def test_gate_is_bit_exact(tag):
    """EffectGate.py:42-126 returns the depth-scaled, envelope-shaped copy (its last line is `return int_array_input`)."""
    kat = load_golden("kat_gate")
    n = 1024
    x = kat["gate_input"]
    keep = x.copy()
    g = ro.OracleGate(**GATE[tag])
    got = np.concatenate([g.apply(x[i * n:(i + 1) * n]) for i in range(16)])
    want = kat["gate_" + tag]
    assert got.dtype == np.float32 and np.array_equal(got, want), int(np.argmax(got != want))
    assert np.array_equal(x, keep)
    # the gate did something besides the plain depth scaling (opened on the loud bursts)
    assert (want != x * np.float32(GATE[tag].get("depth", 0.1))).mean() > 0.2
