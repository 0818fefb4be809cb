#!/usr/bin/env python3
"""Golden vectors for reference CALL PATTERNS that the stream goldens of make_golden.py do not cover (VERDICT r5 "missing" 1, 2, 6).
Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_callpatterns.py

kat_inplace.npz   - Example4's harness (/root/reference/Example4.py:9,18-19): the chunk list becomes ONE 2-D array and every row is
                    overwritten with the device's output, `arr[i] = dev.apply(arr[i])`.  The devices keep VIEWS of the rows as history
                    (EffectFFTFilter.py:63-65 / :139-141, EffectEQ3BandFFT.py:172-174), so call k transforms
                    (out_{k-2}, out_{k-1}, x_k), not (x_{k-2}, x_{k-1}, x_k).  Stored: the array after the loop (what Example4 writes to
                    its WAV file) for LowCut / HighCut / EQ at N = 512 (8 chunks) and N = 88200 (4 chunks, every 64th sample), next to the
                    "clean" stream of the same inputs from separate arrays (Example1's pattern) so that the distance is on record.
kat_nonfinite.npz - one NaN / +Inf / -Inf sample in chunk 2 of 8 (N = 512): per device and value the number of non-finite OUTPUT samples
                    per call.  The reference's 3N-point transform spreads it over the whole output of calls k, k+1, k+2.
Only numbers the reference produced are stored; inputs are regenerated from seeds by the tests.
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = os.environ.get("ADSP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import pyAudioDspTools as ref  # noqa: E402


def stream(seed, n_total):
    return np.random.default_rng(seed).uniform(-1, 1, n_total).astype(np.float32)


DEVICES = {
    "LC": lambda: ref.CreateLowCutFilter(300),
    "HC": lambda: ref.CreateHighCutFilter(8000),
    "EQ": lambda: ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5),
}
SEEDS = {"LC": 201, "HC": 202, "EQ": 203}


def inplace_loop(dev, x):
    """Example4.py:9 and :18-19 with numpy in place of cupy (the cupy twins' apply is the same statements, EffectFFTFilterGPU.py:66-78)."""
    arr = np.array(ref.MakeChunks(x))
    for i in range(len(arr)):
        arr[i] = dev.apply(arr[i])
    return arr


def clean_loop(dev, x):
    return np.stack([dev.apply(c) for c in ref.MakeChunks(x)])


def main():
    kat = {}
    for n, chunks, dec in [(512, 8, 1), (88200, 4, 64)]:
        ref.config.initialize(44100, n)
        for tag, make in DEVICES.items():
            x = stream(SEEDS[tag] + n, chunks * n)
            a = inplace_loop(make(), x.copy())
            c = clean_loop(make(), x.copy())
            assert a.dtype == np.float32 and a.shape == (chunks, n)
            kat[f"{tag}{n}_inplace"] = a.reshape(-1)[::dec].copy()
            kat[f"{tag}{n}_clean"] = c.reshape(-1)[::dec].copy()
            print(f"{tag}{n}: max|inplace - clean| = {np.abs(a - c).max():.4f} at scale {np.abs(c).max():.4f}")
    path = os.path.join(HERE, "kat_inplace.npz")
    np.savez_compressed(path, **kat)
    print("kat_inplace.npz", os.path.getsize(path), "bytes")

    nf = {}
    n, chunks, where = 512, 8, 2 * 512 + 100
    ref.config.initialize(44100, n)
    for tag, make in DEVICES.items():
        for vname, v in [("nan", np.nan), ("pinf", np.inf), ("ninf", -np.inf)]:
            x = stream(SEEDS[tag] + 7, chunks * n)
            x[where] = v
            with np.errstate(all="ignore"):
                out = clean_loop(make(), x)
            nf[f"{tag}_{vname}_nonfinite_per_call"] = (~np.isfinite(out)).sum(axis=1).astype(np.int32)
            nf[f"{tag}_{vname}_nan_per_call"] = np.isnan(out).sum(axis=1).astype(np.int32)
            # the finite calls, to check that the poison leaves nothing behind
            nf[f"{tag}_{vname}_finite_calls"] = out[np.isfinite(out).all(axis=1)].reshape(-1)
            print(tag, vname, nf[f"{tag}_{vname}_nonfinite_per_call"])
    nf["position"] = np.array([where])
    path = os.path.join(HERE, "kat_nonfinite.npz")
    np.savez_compressed(path, **nf)
    print("kat_nonfinite.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
