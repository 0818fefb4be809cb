"""Pin oracle/callers_oracle.py (tapped delay lines behind CreateDelay / the private reverb) to reference goldens."""
import numpy as np
import pytest

from conftest import assert_parity, load_golden, seeded_stream
from oracle import callers_oracle as co

DELAY = {
    # golden name: (fs, chunk, seed, chunks, kwargs)
    "delay4096_default": (44100, 4096, 130, 14, {}),
    "delay4096_wet": (44100, 4096, 130, 14, {"wet": True}),
    "delay4096_300ms_4loops": (44100, 4096, 130, 14, {"time_in_ms": 300, "feedback_loops": 4}),
    "delay512_10ms_5loops": (44100, 512, 131, 12, {"time_in_ms": 10, "feedback_loops": 5}),
    "delay512_7ms_wet": (44100, 512, 131, 12, {"time_in_ms": 7.3, "feedback_loops": 1, "wet": True}),
    "delay512_noloops": (44100, 512, 131, 3, {"time_in_ms": 100, "feedback_loops": 0}),
}


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat_callers")


@pytest.mark.parametrize("name", sorted(DELAY))
def test_delay_oracle_is_bit_exact(kat, name):
    fs, n, seed, chunks, kw = DELAY[name]
    d = co.OracleDelay(fs, n, **kw)
    x = seeded_stream(seed, chunks * n)
    got = np.concatenate([d.apply(x[i * n:(i + 1) * n]) for i in range(chunks)])
    assert got.dtype == np.float32 and np.array_equal(got, kat[name])


@pytest.mark.parametrize("name,fs,n,seed,chunks,ms", [("reverb512_default", 44100, 512, 132, 40, 1500),
                                                      ("reverb256_800ms_48k", 48000, 256, 133, 60, 800)])
def test_reverb_oracle_matches_reference(kat, name, fs, n, seed, chunks, ms):
    rv = co.OracleReverb(fs, n, ms)
    x = seeded_stream(seed, chunks * n)
    got = np.concatenate([rv.applyreverb(x[i * n:(i + 1) * n]) for i in range(chunks)])
    assert_parity(got, kat[name], what=name)
    assert np.abs(kat[name]).max() > 0.05


def test_delay_is_a_sparse_fir(kat):
    """out[t] = x[t] + 0.5 x[t - T] + 0.1 x[t - 2T]: the tap-table form the GPU engine is given."""
    n, chunks = 4096, 14
    x = seeded_stream(130, chunks * n).astype(np.float64)
    taps = co.tap_table(22050, np.linspace(0.5, 0.1, 2, dtype=np.float32))
    assert taps[0][0] == 22050 and taps[1][0] == 44100 and taps[0][1] == 0.5
    want = x.copy()
    for d, g in taps:
        want[d:] += g * x[:-d]
    assert_parity(want, kat["delay4096_default"])


def test_delay_with_filters_is_filter_then_taps():
    """The reference crashes here (EffectDelay.py:56,58 call methods that do not exist); the defined behaviour is its
    reverb delay line's: filter.apply, then the taps."""
    from oracle import fftfilter_oracle as orc
    n = 512
    d = co.OracleDelay(44100, n, 10, 3, 200, 8000, True, True)
    lc, hc, plain = orc.OracleLowCut(200, 44100, n), orc.OracleHighCut(8000, 44100, n), co.OracleDelay(44100, n, 10, 3)
    x = seeded_stream(140, 8 * n)
    for i in range(8):
        ch = x[i * n:(i + 1) * n]
        assert np.array_equal(d.apply(ch), plain.apply(hc.apply(lc.apply(ch))))
