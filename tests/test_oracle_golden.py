"""Pin the CPU oracle (oracle/fftfilter_oracle.py) against vectors captured from the real reference."""
import hashlib

import os

import numpy as np
import pytest

from conftest import assert_parity, seeded_stream
from oracle import fftfilter_oracle as orc

KAT = {
    # name: (factory, fs, N, seed, n_chunks)
    "A": (lambda fs, n: orc.OracleLowCut(800, fs, n), 44100, 4096, 1234, 6),
    "B": (lambda fs, n: orc.OracleHighCut(8000, fs, n), 44100, 4096, 1234, 6),
    "C": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 44100, 512, 1234, 6),
    "D": (lambda fs, n: orc.OracleLowCut(200, fs, n), 44100, 512, 1234, 6),
    "EQ4096": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 44100, 4096, 77, 5),
    "LC8192": (lambda fs, n: orc.OracleLowCut(800, fs, n), 96000, 8192, 78, 4),
    "EQ8192": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 96000, 8192, 79, 4),
    "HC1024": (lambda fs, n: orc.OracleHighCut(20000, fs, n), 48000, 1024, 80, 7),
    "EQ1024": (lambda fs, n: orc.OracleEQ3BandFFT(250, -6, 1500, 3, 6000, -2.5, fs, n), 48000, 1024, 81, 7),
    "LC2048": (lambda fs, n: orc.OracleLowCut(160, fs, n), 44100, 2048, 82, 5),
    "HC256": (lambda fs, n: orc.OracleHighCut(3000, fs, n), 44100, 256, 83, 9),
    "EQ128": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 44100, 128, 84, 9),
    "LC64": (lambda fs, n: orc.OracleLowCut(2000, fs, n), 44100, 64, 85, 11),
    # chunk sizes that are not powers of two
    "LC1000": (lambda fs, n: orc.OracleLowCut(300, fs, n), 44100, 1000, 86, 7),
    "EQ1000": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 44100, 1000, 87, 7),
    "HC1920": (lambda fs, n: orc.OracleHighCut(9000, fs, n), 48000, 1920, 88, 5),
    "LC12000": (lambda fs, n: orc.OracleLowCut(120, fs, n), 44100, 12000, 89, 3),
    "EQ20": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 44100, 20, 90, 40),
    # chunk sizes that are not multiples of 4 (round 4): N // 2 odd gives an EVEN filter length (look-ahead L // 2)
    "LC30": (lambda fs, n: orc.OracleLowCut(3000, fs, n), 44100, 30, 93, 30),
    "HC30": (lambda fs, n: orc.OracleHighCut(8000, fs, n), 44100, 30, 94, 30),
    "EQ30": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 44100, 30, 95, 30),
    "LC1001": (lambda fs, n: orc.OracleLowCut(300, fs, n), 44100, 1001, 96, 7),
    "EQ1001": (lambda fs, n: orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), 44100, 1001, 97, 7),
    "LC1002": (lambda fs, n: orc.OracleLowCut(500, fs, n), 48000, 1002, 98, 7),
    "HC1002": (lambda fs, n: orc.OracleHighCut(9000, fs, n), 48000, 1002, 99, 7),
    "EQ1002": (lambda fs, n: orc.OracleEQ3BandFFT(250, -6, 1500, 3, 6000, -2.5, fs, n), 48000, 1002, 100, 7),
    "HC6": (lambda fs, n: orc.OracleHighCut(8000, fs, n), 44100, 6, 101, 50),
    "LC4410": (lambda fs, n: orc.OracleLowCut(160, fs, n), 44100, 4410, 102, 4),
}
SHA = {"A": "a77f6d09f062", "B": "5af354dacc5b", "C": "9084f1fbd924", "D": "e17160836e0f"}


def run(dev, x, n):
    return np.concatenate([dev.apply(x[i * n:(i + 1) * n]) for i in range(len(x) // n)])


@pytest.mark.parametrize("name", sorted(KAT))
def test_literal_oracle_matches_reference_streams(golden, name):
    make, fs, n, seed, chunks = KAT[name]
    y = run(make(fs, n), seeded_stream(seed, chunks * n), n)
    ref = golden["kat_streams"][name]
    # same arithmetic, same numpy -> bit identical
    assert np.array_equal(y, ref), f"{name}: max diff {np.abs(y - ref).max()}"
    if name in SHA:
        assert hashlib.sha256(y.tobytes()).hexdigest()[:12] == SHA[name]


def test_example4_chunk_size_decimated(golden):
    """N = 88200 (Example4.py:5): the oracle against every 64th sample the reference produced."""
    n = 88200
    y = run(orc.OracleLowCut(300, 44100, n), seeded_stream(91, 3 * n), n)
    assert np.array_equal(y[::64], golden["kat_streams"]["LC88200_dec64"])
    y = run(orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, 44100, n), seeded_stream(92, 3 * n), n)
    assert np.array_equal(y[::64], golden["kat_streams"]["EQ88200_dec64"])


def test_survey_spot_values(golden):
    y = golden["kat_streams"]["A"]
    assert np.allclose(y[2 * 4096:2 * 4096 + 4], [-0.6452144, -0.29364097, 0.22403768, 0.04141868], atol=1e-7)
    assert abs(y[5 * 4096 - 1] - 0.37970021) < 1e-7 and abs(np.abs(y).max() - 1.356733) < 1e-6
    x = seeded_stream(1234, 3)
    assert np.allclose(x, [0.95339954, -0.23960853, 0.84649247])
    h01 = golden["design"]["spot_B_H01"]
    assert abs(h01[0] - 1) < 1e-12 and abs(h01[1] - (0.866280954 - 0.499557113j)) < 1e-8
    h = golden["design"]["spot_A_H0_Hnyq"]
    assert abs(h[0]) < 1e-12 and abs(abs(h[1]) - 1) < 1e-9


def test_design_kernels_match_reference(golden):
    g = golden["design"]
    for key in g.files:
        parts = key.split("_")
        if parts[0] == "lowcut" and parts[1] != "default":
            fs, n, fc = int(parts[1]), int(parts[2]), float(parts[3])
            mine = orc.lowcut_taps(fc, fs, n)
        elif parts[0] == "highcut" and parts[1] != "default":
            fs, n, fc = int(parts[1]), int(parts[2]), float(parts[3])
            mine = orc.highcut_taps(fc, fs, n)
        elif parts[0] == "eq":
            fs, n = int(parts[1]), int(parts[2])
            p = [float(v) for v in parts[3:9]]
            band = "_".join(parts[9:])
            mine = orc.eq3_band_taps(p[0], p[2], p[4], fs, n)[band]
        elif key == "highcut_default_44100_512":
            mine = orc.highcut_taps(8000, 44100, 512)
        elif key == "lowcut_default_44100_512":
            mine = orc.lowcut_taps(160, 44100, 512)
        else:
            continue
        assert np.abs(mine - g[key]).max() < 1e-12, key


def test_chain_E(golden):
    n, fs = 8192, 96000
    a, b, c = orc.OracleLowCut(800, fs, n), orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), orc.OracleHighCut(8000, fs, n)
    x = seeded_stream(4321, 12 * n)
    y = np.concatenate([c.apply(b.apply(a.apply(x[i * n:(i + 1) * n]))) for i in range(12)])
    assert np.array_equal(y, golden["kat_chain"]["E"])


def test_example1_plumbing_F(golden):
    g = golden["kat_example1"]
    x = g["pcm16_first8"].astype(np.float32) / 32768  # Utility.py:236-238
    dev = orc.OracleLowCut(800, 44100, 4096)
    y = run(dev, x, 4096)
    assert np.array_equal(y, g["out_first8"])
    # MakeChunks quirk: 264600 samples -> 65 chunks, padded to 266240
    chunks = orc.make_chunks(np.zeros(264600, np.float32), 4096)
    assert len(chunks) == 65 and sum(len(c) for c in chunks) == int(g["out_len"][0]) == 266240
    assert len(orc.combine_chunks(chunks)) == 266240


def test_example1_full_run_hash():
    """the whole of Example1.py (65 chunks, 1640 padded zeros, last input chunk never flushed) through the
    oracle reproduces the reference's merged output - bit for bit (sha256 prefix 9c38cf3169419998) under the numpy the
    fixture was made with, to 1e-6 of full scale under any other."""
    import hashlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_example1_full.npz"))
    full = g["pcm16"].astype(np.float32) / 32768
    chunks = orc.make_chunks(full, 4096)
    assert len(chunks) == int(g["n_chunks"][0]) == 65 and not chunks[-1][-1640:].any() and len(chunks[-1]) == 4096
    dev = orc.OracleLowCut(800, 44100, 4096)
    outs = [dev.apply(c) for c in chunks]
    merged = orc.combine_chunks(outs)
    assert len(merged) == int(g["out_len"][0]) == 266240
    assert np.abs(merged[::8] - g["out_dec8"]).max() <= 1e-6
    assert np.abs(outs[0] - g["out_first_chunk"]).max() <= 1e-6 and np.abs(outs[-1] - g["out_last_chunk"]).max() <= 1e-6
    if np.__version__ == bytes(g["numpy_version"]).decode():
        assert hashlib.sha256(merged.astype(np.float32).tobytes()).hexdigest()[:16] == bytes(g["sha16"]).decode() == "9c38cf3169419998"


EDGE_INPUTS = {
    "zeros": lambda n: np.zeros(5 * n, np.float32),
    "imp0": lambda n: np.eye(1, 5 * n, 0, dtype=np.float32)[0],
    "impNm1": lambda n: np.eye(1, 5 * n, n - 1, dtype=np.float32)[0],
    "impN": lambda n: np.eye(1, 5 * n, n, dtype=np.float32)[0],
    "dc": lambda n: np.ones(5 * n, np.float32),
    "square": lambda n: np.where((np.arange(5 * n) // 37) % 2 == 0, 1.0, -1.0).astype(np.float32),
}


@pytest.mark.parametrize("edge", sorted(EDGE_INPUTS))
def test_edges_G(golden, edge):
    n = 512
    x = EDGE_INPUTS[edge](n)
    g = golden["kat_edges"]
    assert np.array_equal(run(orc.OracleLowCut(200, 44100, n), x, n), g["lowcut_" + edge])
    assert np.array_equal(run(orc.OracleHighCut(8000, 44100, n), x, n), g["highcut_" + edge])
    assert np.array_equal(run(orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, 44100, n), x, n), g["eq_" + edge])


def test_input_types(golden):
    n = 512
    x64 = np.random.default_rng(5).uniform(-1, 1, 4 * n)
    g = golden["kat_edges"]
    assert np.array_equal(run(orc.OracleLowCut(200, 44100, n), x64, n), g["lowcut_f64in"])
    dev = orc.OracleHighCut(8000, 44100, n)
    y = np.concatenate([dev.apply(list(x64[i * n:(i + 1) * n])) for i in range(4)])
    assert np.array_equal(y, g["highcut_listin"])


# ---- independent ground truth: float64 direct convolution (no FFT) -----------------------------
@pytest.mark.parametrize("name", ["A", "B", "C", "D", "HC256", "EQ128", "LC64", "EQ1024", "LC1000", "EQ1000", "EQ20",
                                  "LC30", "HC30", "EQ30", "LC1001", "EQ1001", "LC1002", "HC1002", "EQ1002", "HC6", "LC4410"])
def test_direct_convolution_identity(golden, name):
    make, fs, n, seed, chunks = KAT[name]
    x = seeded_stream(seed, chunks * n)
    dev = make(fs, n)
    if isinstance(dev, orc.OracleEQ3BandFFT):
        params = {"EQ1024": (250, -6, 1500, 3, 6000, -2.5), "EQ1002": (250, -6, 1500, 3, 6000, -2.5)}.get(name, (100, 2, 700, -4, 8000, 5))
        taps = orc.eq3_composite_taps(*params, fs, n)
    else:
        taps = np.fft.ifft(dev.spectrum).real[: n // 2 - 1]
    truth = orc.direct_stream_convolution(taps, x, n)
    assert_parity(golden["kat_streams"][name], truth, what=name)


def test_chain_is_one_fir(golden):
    """LowCut -> EQ3 -> HighCut == one (4L-3)-tap FIR with 3 chunks latency."""
    n, fs = 8192, 96000
    lc, hc = orc.lowcut_taps(800, fs, n), orc.highcut_taps(8000, fs, n)
    eq = orc.eq3_composite_taps(100, 2, 700, -4, 8000, 5, fs, n)
    comp = np.convolve(np.convolve(lc, eq), hc)
    d = orc.geometry(n)[1]
    x = seeded_stream(4321, 12 * n)
    truth = orc.direct_stream_convolution(comp, x, n, latency_chunks=3, lookahead=3 * d)
    assert_parity(golden["kat_chain"]["E"], truth, what="chain")


@pytest.mark.parametrize("name", ["A", "C", "EQ4096", "HC1024", "LC64"])
def test_rfft2n_formulation_matches(golden, name):
    """The 2N real-FFT overlap-save restatement (what the GPU computes) is inside tolerance."""
    make, fs, n, seed, chunks = KAT[name]
    dev = make(fs, n)
    if isinstance(dev, orc.OracleEQ3BandFFT):
        taps = orc.eq3_composite_taps(*({"C": (100, 2, 700, -4, 8000, 5), "EQ4096": (100, 2, 700, -4, 8000, 5)}[name]), fs, n)
    else:
        taps = np.fft.ifft(dev.spectrum).real[: n // 2 - 1]
    x = seeded_stream(seed, chunks * n)
    fast = orc.OracleRfft2N(taps, n, channels=1)
    y = np.concatenate([fast.apply(x[i * n:(i + 1) * n][None, :])[0] for i in range(chunks)])
    assert_parity(y, golden["kat_streams"][name], what=name)
