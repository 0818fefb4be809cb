"""CPU tests of the chunk / WAV plumbing: reference semantics of the host helpers."""
import os
import wave

import numpy as np
import pytest

from conftest import seeded_stream
from oracle import fftfilter_oracle as orc


def _write_wav(path, int16_frames, n_channels, rate=44100):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(n_channels)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.ascontiguousarray(int16_frames, dtype=np.int16).tobytes())


def test_wav_import_matches_reference_semantics(tmp_path):
    import pyaudiodsptools_amd as adsp
    rng = np.random.default_rng(0)
    mono = rng.integers(-32768, 32768, 1000, dtype=np.int16)
    _write_wav(tmp_path / "m.wav", mono, 1)
    assert np.array_equal(adsp.MonoWavToNumpy16BitInt(str(tmp_path / "m.wav")), mono)
    f = adsp.MonoWavToNumpyFloat(str(tmp_path / "m.wav"))
    assert f.dtype == np.float32 and np.array_equal(f, orc.pcm16_to_float(mono))
    st = rng.integers(-32768, 32768, (500, 2), dtype=np.int16)
    _write_wav(tmp_path / "s.wav", st, 2)
    left, right = adsp.StereoWavToNumpyFloat(str(tmp_path / "s.wav"))
    assert np.array_equal(left, st[:, 0].astype(np.float32) / 32768) and np.array_equal(right, st[:, 1].astype(np.float32) / 32768)
    with pytest.raises(ValueError):
        adsp.StereoWavToNumpyFloat(str(tmp_path / "m.wav"))


def test_wav_export_truncates_like_the_reference(tmp_path):
    import pyaudiodsptools_amd as adsp
    adsp.config.initialize(48000, 512)
    x = np.array([0.0, 0.5, -0.5, 0.99999, -0.99999, 1.0, -1.0, 3.05e-5, -3.05e-5, 0.25001], np.float32)
    adsp.NumpyFloatToWav(str(tmp_path / "o.wav"), x)
    with wave.open(str(tmp_path / "o.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 48000)
        got = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    assert np.array_equal(got, orc.float_to_pcm16(x)) and got[7] == 0 and got[8] == 0  # truncation, not rounding
    stereo = np.stack([x, -x])  # (2, n) is transposed to (n, 2) like Utility.py:286-287
    adsp.NumpyFloatToWav(str(tmp_path / "o2.wav"), stereo)
    with wave.open(str(tmp_path / "o2.wav")) as w:
        assert w.getnchannels() == 2
        got = np.frombuffer(w.readframes(w.getnframes()), np.int16).reshape(-1, 2)
    assert np.array_equal(got[:, 0], orc.float_to_pcm16(x)) and np.array_equal(got[:, 1], orc.float_to_pcm16(-x))
    with pytest.raises(ValueError):
        adsp.NumpyFloatToWav(str(tmp_path / "bad.wav"), np.array([2.0, 3.0]))


def test_chunk_plumbing_matches_reference_quirks():
    import pyaudiodsptools_amd as adsp
    adsp.config.initialize(44100, 4096)
    x = seeded_stream(3, 264600)
    mine, ref = adsp.MakeChunks(x), orc.make_chunks(x, 4096)
    assert len(mine) == len(ref) == 65 and all(np.array_equal(a, b) for a, b in zip(mine, ref))
    assert np.array_equal(adsp.CombineChunks(mine), orc.combine_chunks(ref))
    # a length that is a multiple of the chunk count but not of the chunk size is NOT padded (Utility.py:23)
    y = seeded_stream(4, 3 * 4000)
    assert [len(c) for c in adsp.MakeChunks(y)] == [len(c) for c in orc.make_chunks(y, 4096)] == [4000, 4000, 4000]
    assert adsp.CombineChunks([]).dtype == np.float32


def test_oracle_pcm_pipeline_matches_golden_example1_and_2(golden):
    g1 = golden["kat_example1"]
    got = orc.run_device_pcm16(orc.OracleLowCut(800, 44100, 4096), g1["pcm16_first8"], 4096)
    assert np.array_equal(got, orc.float_to_pcm16(g1["out_first8"]))
    g2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_example2.npz"))
    pcm = g2["pcm16_first4_stereo"]
    for ch, key in ((0, "out_left"), (1, "out_right")):
        dev = orc.OracleLowCut(800, 44100, 4096)
        x = orc.pcm16_to_float(pcm[:, ch])
        y = np.concatenate([dev.apply(x[i * 4096:(i + 1) * 4096]) for i in range(4)])
        assert np.array_equal(y, g2[key])


def test_wavbank_layout(tmp_path):
    import pyaudiodsptools_amd as adsp
    rng = np.random.default_rng(1)
    a = rng.integers(-3000, 3000, 1300, dtype=np.int16)
    b = rng.integers(-3000, 3000, (700, 2), dtype=np.int16)
    _write_wav(tmp_path / "a.wav", a, 1)
    _write_wav(tmp_path / "b.wav", b, 2, rate=48000)
    bank = adsp.WavBank([str(tmp_path / "a.wav"), str(tmp_path / "b.wav")], chunk_size=512)
    assert (bank.channels, bank.steps) == (3, 3)
    batch = bank.batch()
    assert batch.shape == (3, 3, 512) and batch.dtype == np.int16
    assert np.array_equal(batch[:, 0].reshape(-1)[:1300], a) and not batch[:, 0].reshape(-1)[1300:].any()
    assert np.array_equal(batch[:, 2].reshape(-1)[:700], b[:, 1])
