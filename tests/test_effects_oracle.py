"""Pin oracle/effects_oracle.py against vectors captured from the reference's wave-shapers (tests/golden/kat_effects.npz)."""
import numpy as np
import pytest

from conftest import assert_parity, seeded_stream
from oracle import effects_oracle as fx
from oracle import fftfilter_oracle as orc

LOUD = lambda: (seeded_stream(100, 4096) * np.float32(1.5)).astype(np.float32)  # noqa: E731

STANDALONE = {
    "softclip_044": lambda x: fx.soft_clipper(x),
    "softclip_200": lambda x: fx.soft_clipper(x, 2.0),
    "harddist": fx.hard_distortion,
    "saturator_hard": lambda x: fx.saturator(x),
    "saturator_soft": lambda x: fx.saturator(x, -12.0, 3.0, "soft"),
    "volume_p6_clip": lambda x: fx.volume_change(x, 6.0),
    "volume_m35_noclip": lambda x: fx.volume_change(x, -3.5, False),
}


@pytest.mark.parametrize("name", sorted(STANDALONE))
def test_effect_oracle_matches_reference(golden, name):
    want = golden["kat_effects"][name]
    got = STANDALONE[name](LOUD())
    assert want.dtype == np.float32 and got.dtype == np.float32
    # same float32 operations in the same order: at most an ulp or two apart (numpy's pow/sin are the same code)
    assert np.abs(got.astype(np.float64) - want).max() <= 2.5e-7, name


def test_bit_crusher_oracle_matches_reference(golden):
    want = golden["kat_effects"]["bitcrusher"]
    got = fx.bit_crusher(seeded_stream(100, 4096))
    assert want.dtype == np.float64 and np.array_equal(got, want)
    assert np.array_equal(want * 64, np.round(want * 64)) and np.abs(want).max() <= 1.0


def test_hard_distortion_quirks_are_the_references(golden):
    """0 maps to +0.951 (not 0) and loud negative samples to -0.718 (not -0.968): documented reference behaviour."""
    y = fx.hard_distortion(np.array([0.0, 2.0, -2.0, 0.8, -0.8], np.float32))
    assert_parity(y, [0.8 + 0.2 * np.sin(-4.0), 0.8 + 0.2 * np.sin(1.0), -(0.8 + 0.2 * np.sin(-9.0)), 0.8, -0.8])


CHAINS = {
    "chain512_lowcut_softclip": (512, 101, 8, lambda: orc.OracleLowCut(200, 44100, 512), lambda y: fx.soft_clipper(y)),
    "chain512_eq_saturator_soft": (512, 101, 8, lambda: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, 44100, 512),
                                   lambda y: fx.saturator(y, -12.0, 3.0, "soft")),
    "chain4096_lowcut_saturator_hard": (4096, 102, 5, lambda: orc.OracleLowCut(800, 44100, 4096), lambda y: fx.saturator(y)),
    "chain4096_eq_volume_p3": (4096, 102, 5, lambda: orc.OracleEQ3BandFFT(100, 6, 700, 3, 8000, 6, 44100, 4096),
                               lambda y: fx.volume_change(y, 3.0)),
}


@pytest.mark.parametrize("name", sorted(CHAINS))
def test_filter_then_effect_oracle_matches_reference(golden, name):
    n, seed, chunks, make, effect = CHAINS[name]
    dev = make()
    x = seeded_stream(seed, chunks * n)
    got = np.concatenate([effect(dev.apply(x[i * n:(i + 1) * n])) for i in range(chunks)])
    assert_parity(got, golden["kat_effects"][name], what=name)


def test_filter_then_hard_distortion_oracle(golden):
    """The distortion is discontinuous at 0 and |x| = 0.8: compare away from those points."""
    n, chunks = 512, 8
    dev = orc.OracleHighCut(8000, 44100, n)
    x = seeded_stream(101, chunks * n)
    pre = np.concatenate([dev.apply(x[i * n:(i + 1) * n]) for i in range(chunks)])
    safe = (np.abs(pre) > 1e-5) & (np.abs(np.abs(pre) - 0.8) > 1e-5) | (pre == 0)
    assert safe.mean() > 0.9  # the first chunk is mostly round-off noise around 0, whose sign the distortion amplifies
    assert_parity(fx.hard_distortion(pre)[safe], golden["kat_effects"]["chain512_highcut_harddist"][safe])


TREMOLO = {
    # golden name: (fs, chunk, seed, chunks, depth, lfo)
    "tremolo_default": (44100, 4096, 103, 6, 0.4, 4.5),
    "tremolo_48k_7hz": (48000, 1000, 104, 12, 0.9, 7),
    "tremolo_quirk": (44100, 512, 105, 8, 0.5, 44100 / 1536),
}


@pytest.mark.parametrize("name", sorted(TREMOLO))
def test_tremolo_oracle_matches_reference(golden, name):
    fs, n, seed, chunks, depth, lfo = TREMOLO[name]
    t = fx.OracleTremolo(fs, depth, lfo)
    x = seeded_stream(seed, chunks * n)
    got = np.concatenate([t.apply(x[i * n:(i + 1) * n]) for i in range(chunks)])
    assert np.array_equal(got, golden["kat_effects"][name]), name  # same float32 table, same products
    if name == "tremolo_48k_7hz":
        assert len(t.table) == int(golden["kat_effects"]["tremolo_48k_7hz_len"][0]) == 6858


def test_tremolo_quirk_replays_one_segment(golden):
    """Period 1536 = 3 chunks of 512: chunks 0, 1, 2 walk the table, chunk 2 is then replayed for ever."""
    t = fx.OracleTremolo(44100, 0.5, 44100 / 1536)
    ones = np.ones(512, np.float32)
    g = [t.apply(ones) for _ in range(6)]
    assert not np.array_equal(g[0], g[1]) and not np.array_equal(g[1], g[2])
    assert all(np.array_equal(g[2], g[k]) for k in (3, 4, 5))
    t.reset()
    assert np.array_equal(t.apply(ones), g[0])


def test_tremolo_chain_and_mix_oracles(golden):
    n = 512
    x = seeded_stream(106, 12 * n)
    for name, make_dev, depth, lfo in [("chain512_lowcut_tremolo", lambda: orc.OracleLowCut(200, 44100, n), 0.6, 10),
                                       ("chain512_highcut_tremolo_quirk", lambda: orc.OracleHighCut(8000, 44100, n), 0.5, 44100 / 1536)]:
        dev, t = make_dev(), fx.OracleTremolo(44100, depth, lfo)
        got = np.concatenate([t.apply(dev.apply(x[i * n:(i + 1) * n])) for i in range(12)])
        assert_parity(got, golden["kat_effects"][name], what=name)
    a, b, c = seeded_stream(107, 4096), seeded_stream(108, 4096), seeded_stream(109, 4096)
    want = golden["kat_effects"]["mix3"]
    assert want.dtype == np.float64 and np.array_equal(fx.mix_signals(a, b, c), want)
    assert (np.abs(want) == 1.0).mean() > 0.05  # the clip is exercised
    d = [orc.OracleLowCut(200, 44100, n), orc.OracleHighCut(8000, 44100, n), orc.OracleEQ3BandFFT(100, 6, 700, 3, 8000, 6, 44100, n)]
    xs = [seeded_stream(110 + k, 6 * n) for k in range(3)]
    got = np.concatenate([fx.mix_signals(*[d[k].apply(xs[k][i * n:(i + 1) * n]) for k in range(3)]) for i in range(6)])
    assert_parity(got, golden["kat_effects"]["chain512_mix3"], what="mix chain")
