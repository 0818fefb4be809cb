"""The oracle against the call-pattern goldens (tests/golden/make_golden_callpatterns.py, captured from the real reference):
Example4's in-place loop (the device's history ALIASES rows the caller overwrites with outputs) and non-finite samples."""
import numpy as np
import pytest

from conftest import load_golden, seeded_stream
from oracle import fftfilter_oracle as orc

DEVICES = {
    "LC": lambda fs, n: orc.OracleLowCut(300, fs, n),
    "HC": lambda fs, n: orc.OracleHighCut(8000, fs, n),
    "EQ": lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n),
}
SEEDS = {"LC": 201, "HC": 202, "EQ": 203}


def inplace_loop(dev, x, n):
    """Example4.py:9,18-19: ONE 2-D array, every row overwritten with the output of the call that took it."""
    arr = np.array(orc.make_chunks(x, n))
    for i in range(len(arr)):
        arr[i] = dev.apply(arr[i])
    return arr


@pytest.mark.parametrize("tag", ["LC", "HC", "EQ"])
@pytest.mark.parametrize("n,chunks,dec", [(512, 8, 1), (88200, 4, 64)])
def test_oracle_reproduces_example4s_in_place_loop_bit_for_bit(tag, n, chunks, dec):
    kat = load_golden("kat_inplace")
    x = seeded_stream(SEEDS[tag] + n, chunks * n)
    got = inplace_loop(DEVICES[tag](44100, n), x.copy(), n).reshape(-1)[::dec]
    assert np.array_equal(got, kat[f"{tag}{n}_inplace"])
    clean = np.concatenate([c for c in map(DEVICES[tag](44100, n).apply, orc.make_chunks(x.copy(), n))])[::dec]
    assert np.array_equal(clean, kat[f"{tag}{n}_clean"])
    # the two streams are full-scale apart: the in-place loop is NOT the FIR stream of its input
    assert np.abs(kat[f"{tag}{n}_inplace"] - kat[f"{tag}{n}_clean"]).max() > 0.5 * np.abs(kat[f"{tag}{n}_clean"]).max()


def test_in_place_loop_is_the_filter_over_previous_outputs():
    """What the aliasing means: call k transforms (out[k-2], out[k-1], x[k]) - the statement alias_history=True is built on."""
    n, chunks = 512, 8
    kat = load_golden("kat_inplace")
    for tag in DEVICES:
        x = seeded_stream(SEEDS[tag] + n, chunks * n).reshape(chunks, n)
        outs, zero = [], np.zeros(n, np.float32)
        for k in range(chunks):
            dev = DEVICES[tag](44100, n)
            dev.apply(outs[k - 2] if k >= 2 else zero)
            dev.apply(outs[k - 1] if k >= 1 else zero)
            outs.append(dev.apply(x[k]))
        want = kat[f"{tag}{n}_inplace"].reshape(chunks, n)
        assert np.abs(np.stack(outs) - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("tag", ["LC", "HC", "EQ"])
@pytest.mark.parametrize("vname,value", [("nan", np.nan), ("pinf", np.inf), ("ninf", -np.inf)])
def test_oracle_non_finite_sample_poisons_three_whole_calls(tag, vname, value):
    nf = load_golden("kat_nonfinite")
    n, chunks, where = 512, 8, int(nf["position"][0])
    x = seeded_stream(SEEDS[tag] + 7, chunks * n)
    x[where] = value
    dev = DEVICES[tag](44100, n)
    with np.errstate(all="ignore"):
        out = np.stack([dev.apply(c) for c in orc.make_chunks(x, n)])
    per_call = (~np.isfinite(out)).sum(axis=1)
    assert np.array_equal(per_call, nf[f"{tag}_{vname}_nonfinite_per_call"])
    assert np.array_equal(per_call, [0, 0, n, n, n, 0, 0, 0])  # the call that takes the sample and the two after it, every sample
    assert np.array_equal(np.isnan(out).sum(axis=1), nf[f"{tag}_{vname}_nan_per_call"])
    assert np.array_equal(out[np.isfinite(out).all(axis=1)].reshape(-1), nf[f"{tag}_{vname}_finite_calls"])
