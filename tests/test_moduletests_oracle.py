"""The reference's own harness, ModuleTests.py, restated with the CPU oracle and pinned against what the reference computes in it
(tests/golden/kat_moduletests.npz, made by tests/golden/make_golden_moduletests.py): a 1 kHz sine at 44100 Hz / 512 samples goes
through Saturator -> Compressor -> Delay -> Tremolo -> HardDistortion -> Gate -> LowCut(200) -> HighCut(8000) -> EQ3BandFFT -> SoftClipper,
every device created with the arguments of ModuleTests.py:73-84 and fed the previous device's output chunk by chunk (:95-214).
The GPU twin of this test is tests/test_gpu_moduletests.py."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import callers_oracle, effects_oracle as fx, fftfilter_oracle as orc, recursive_oracle as rec

FS, N = 44100, 512


class _Stateless:
    def __init__(self, fn):
        self.apply = fn


def oracle_devices():
    """ModuleTests.py:73-84 in the order its loops use them (:95-214)."""
    return [_Stateless(fx.saturator), rec.OracleCompressor(FS), callers_oracle.OracleDelay(FS, N), fx.OracleTremolo(FS),
            _Stateless(fx.hard_distortion), rec.OracleGate(), orc.OracleLowCut(200, FS, N), orc.OracleHighCut(8000, FS, N),
            orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, FS, N), _Stateless(fx.soft_clipper)]


def run_chunks(dev, stream):
    return np.concatenate([np.asarray(dev.apply(c.copy())) for c in stream.reshape(-1, N)])


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat_moduletests")


def test_generators_of_the_harness(kat):
    """ModuleTests.py:57-59: CreateSinewave / CreateSquarewave (Generators.py:25-27, :51-55) as float64 expressions rounded once."""
    t = np.arange(int(kat["length"]))
    sine = np.float32(np.sin(2 * np.pi * 1000 * t / FS))
    assert np.array_equal(sine, kat["sine"])
    assert np.array_equal(np.where(sine > 0, 1, -1).astype(np.int8), kat["square"]) and str(kat["square_dtype"]) == "float64"
    chunked = np.concatenate([sine, np.zeros(-len(sine) % N, np.float32)])  # MakeChunks pads with zeros (Utility.py:22-28)
    assert np.array_equal(chunked, kat["stage_00"])


def test_each_device_on_the_references_own_input(kat):
    """Stage k of the harness in isolation: the oracle's device k on what the REFERENCE handed its device k."""
    names = [str(s) for s in kat["stage_names"]]
    for k, dev in enumerate(oracle_devices(), start=1):
        got = run_chunks(dev, kat[f"stage_{k - 1:02d}"])
        want = kat[f"stage_{k:02d}"]
        assert str(kat[f"stage_{k:02d}_dtype"]) == "float32"
        err = np.abs(got.astype(np.float64) - want).max()
        # float32 expressions in the reference's order: an ulp or two (numpy's pow / sin may differ by one between builds)
        assert err <= 3e-7 * max(1.0, np.abs(want).max()), (names[k - 1], err)


def test_the_whole_harness_end_to_end(kat):
    """All ten loops chained on the oracle's own intermediate results, like ModuleTests.py does: the end of the chain (what it passes to
    CombineChunks, :217) matches the reference's."""
    stream = kat["stage_00"]
    for k, dev in enumerate(oracle_devices(), start=1):
        stream = run_chunks(dev, stream)
        want = kat[f"stage_{k:02d}"]
        assert np.abs(stream.astype(np.float64) - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), str(kat["stage_names"][k - 1])
    assert stream.dtype == np.float32 and np.abs(stream).max() <= 1.0  # the soft clipper ends the chain inside [-1, 1]


def test_the_gpu_harness_in_place_loops(kat):
    """ModuleTestsGPU.py:35-110: chunk 88200, ONE 2-D array, three devices in a row, every loop `arr[i] = dev.apply(arr[i])`.  The devices
    keep views of the rows (EffectFFTFilterGPU.py:66-68), so each loop filters its own previous outputs - the oracle keeps references like
    the reference and reproduces both the in-place arrays and the clean streams."""
    chunks, n, dec = (int(v) for v in kat["gpu_shape"])
    sine = np.float32(np.sin(2 * np.pi * 1000 * np.arange(chunks * n) / FS))
    make = [lambda: orc.OracleLowCut(200, FS, n), lambda: orc.OracleHighCut(8000, FS, n),
            lambda: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, FS, n)]
    arr = sine.reshape(chunks, n).copy()
    clean = [c.copy() for c in sine.reshape(chunks, n)]
    for k, mk in enumerate(make, start=1):
        dev = mk()
        for i in range(len(arr)):
            arr[i] = dev.apply(arr[i])
        assert np.array_equal(arr.reshape(-1)[::dec], kat[f"gpu_inplace_{k}"]), k
        dev = mk()
        clean = [dev.apply(c) for c in clean]
        assert np.array_equal(np.concatenate(clean)[::dec], kat[f"gpu_clean_{k}"]), k
    # what the aliasing does to the script's signal: a 1 kHz sine, which all three devices pass, comes out 220 dB down
    assert np.abs(kat["gpu_clean_3"]).max() > 1.0 and np.abs(kat["gpu_inplace_3"]).max() < 1e-10
