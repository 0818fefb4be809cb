"""Host side of the reference's remaining exports (pyaudiodsptools_amd/signals.py) and of compat.install(): pinned against
tests/golden/kat_moduletests.npz (values the reference produced) - no GPU, no compute call into libadsp."""
import sys

import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture()
def kat():
    from pyaudiodsptools_amd import config
    keep = (config.sampling_rate, config.chunk_size)
    config.initialize(44100, 512)
    yield load_golden("kat_moduletests")
    config.sampling_rate, config.chunk_size = keep


def test_sine_and_square_are_the_references(kat):
    import pyaudiodsptools_amd as adsp
    n = int(kat["length"])
    sine, square = adsp.CreateSinewave(1000, n), adsp.CreateSquarewave(1000, n)
    assert sine.dtype == np.float32 and np.array_equal(sine, kat["sine"])
    assert str(square.dtype) == str(kat["square_dtype"]) and np.array_equal(square.astype(np.int8), kat["square"])
    assert set(np.unique(square)) == {-1.0, 1.0}
    assert adsp.CreateSinewave(440, 0).shape == (0,)


def test_white_noise_has_the_references_spectrum(kat):
    import pyaudiodsptools_amd as adsp
    n = int(kat["length"])
    for length in (n, n + 1):  # even and odd lengths (the Nyquist bin exists for even n only)
        noise = adsp.CreateWhitenoise(length, seed=5)
        assert str(noise.dtype) == str(kat["noise_dtype"]) and noise.shape == (length,)
        mag = np.abs(np.fft.rfft(noise.astype(np.float64))) / 5
        freqs = np.fft.rfftfreq(length, 1 / 44100)
        inband = (freqs >= 20) & (freqs <= 20000)
        assert abs(mag[inband].min() - 1) < 1e-6 and abs(mag[inband].max() - 1) < 1e-6 and mag[~inband].max() < 1e-6
        if length == n:
            assert list(np.flatnonzero(inband)[[0, -1]]) == list(kat["noise_inband_bins"])
            assert abs(np.sqrt(np.mean(noise.astype(np.float64) ** 2)) - float(kat["noise_rms"])) < 1e-9
            lo, hi = kat["noise_inband_mag_minmax"]
            assert abs(lo - 1) < 1e-6 and abs(hi - 1) < 1e-6 and float(kat["noise_outband_mag_max"]) < 1e-6  # the reference's own
    assert np.array_equal(adsp.CreateWhitenoise(4096, seed=9), adsp.CreateWhitenoise(4096, seed=9))
    assert not np.array_equal(adsp.CreateWhitenoise(4096), adsp.CreateWhitenoise(4096))  # unseeded like the reference's


def test_level_helpers_are_the_references(kat):
    import pyaudiodsptools_amd as adsp
    sine = kat["sine"]
    as16 = adsp.ConvertdBVTo16Bit(sine * 1.5)
    assert str(as16.dtype) == str(kat["to16_dtype"]) and np.array_equal(as16, kat["to16"])
    assert as16.max() == 32767 and as16.min() == -32767  # clipped at one volt, not wrapped
    back = adsp.Convert16BitTodBV(as16)
    assert str(back.dtype) == str(kat["from16_dtype"]) and np.array_equal(back, kat["from16"])
    assert adsp.InfodBV(sine) == float(kat["info_dbv"]) and adsp.InfodBV16Bit(as16) == float(kat["info_db16"])
    with pytest.raises(ValueError):
        adsp.InfodBV(np.zeros(8, np.float32))  # math.log10(0), like the reference


def test_dither_is_zero_or_minus_one_lsb(kat):
    import pyaudiodsptools_amd as adsp
    rng = np.random.default_rng(77)
    i16 = rng.integers(-32768, 32768, 4096).astype(np.int16)
    i32 = rng.integers(-2 ** 31, 2 ** 31, 4096).astype(np.int64)
    d8, d16 = adsp.Dither16BitTo8Bit(i16), adsp.Dither32BitIntTo16BitInt(i32, rng=np.random.default_rng(3))
    assert str(d8.dtype) == str(kat["dither8_dtype"]) and str(d16.dtype) == str(kat["dither16_dtype"])
    assert np.array_equal(np.unique(np.clip(np.around(i16 / 256), -127, 127) - d8), kat["dither8_offsets"])
    assert np.array_equal(np.unique(np.clip(np.around(i32 / 65535), -32767, 32767) - d16), kat["dither16_offsets"])
    assert np.abs(d8).max() <= 127 and np.abs(d16.astype(np.int64)).max() <= 32767
    assert np.array_equal(adsp.Dither32BitIntTo16BitInt(i32, rng=np.random.default_rng(3)), d16)


def test_compat_registers_the_references_module_layout():
    """Every import line of ModuleTests.py:12, :36-52 resolves to objects of this package; uninstall removes the names again."""
    import pyaudiodsptools_amd as adsp
    from pyaudiodsptools_amd import compat
    assert "pyAudioDspTools" not in sys.modules
    compat.install()
    try:
        import pyAudioDspTools
        from pyAudioDspTools import config
        from pyAudioDspTools.Generators import CreateSinewave, CreateSquarewave, CreateWhitenoise
        from pyAudioDspTools.Utility import MakeChunks, CombineChunks, MixSignals, ConvertdBVTo16Bit
        from pyAudioDspTools.Utility import Convert16BitTodBV, Dither16BitTo8Bit, Dither32BitIntTo16BitInt, MonoWavToNumpyFloat, InfodBV
        from pyAudioDspTools.Utility import InfodBV16Bit, VolumeChange, MonoWavToNumpy16BitInt, NumpyFloatToWav
        from pyAudioDspTools.EffectCompressor import CreateCompressor
        from pyAudioDspTools.EffectGate import CreateGate
        from pyAudioDspTools.EffectDelay import CreateDelay
        from pyAudioDspTools._EffectReverb import CreateReverb
        from pyAudioDspTools.EffectFFTFilter import CreateHighCutFilter, CreateLowCutFilter
        from pyAudioDspTools.EffectEQ3BandFFT import CreateEQ3BandFFT
        from pyAudioDspTools.EffectEQ3Band import CreateEQ3Band
        from pyAudioDspTools.EffectSoftClipper import CreateSoftClipper
        from pyAudioDspTools.EffectHardDistortion import CreateHardDistortion
        from pyAudioDspTools.EffectTremolo import CreateTremolo
        from pyAudioDspTools.EffectSaturator import CreateSaturator
        from pyAudioDspTools.EffectFFTFilterGPU import CreateHighCutFilterGPU, CreateLowCutFilterGPU
        from pyAudioDspTools.EffectEQ3BandFFTGPU import CreateEQ3BandFFTGPU
        assert pyAudioDspTools is adsp and config is adsp.config
        got = dict(locals())
        for module_name, names in compat.LAYOUT.items():
            for n in names:
                assert getattr(sys.modules["pyAudioDspTools." + module_name], n) is getattr(adsp, n), (module_name, n)
                if n in got:
                    assert got[n] is getattr(adsp, n), n
        # every name the reference's package exports (pyAudioDspTools/__init__.py:11-28) exists at the top level too
        for n in ("CreateSinewave CreateSquarewave CreateWhitenoise MakeChunks CombineChunks MixSignals ConvertdBVTo16Bit Convert16BitTodBV "
                  "Dither16BitTo8Bit Dither32BitIntTo16BitInt MonoWavToNumpyFloat InfodBV InfodBV16Bit VolumeChange MonoWavToNumpy16BitInt "
                  "NumpyFloatToWav CreateCompressor CreateSoftClipper CreateSaturator CreateGate CreateDelay CreateHighCutFilter "
                  "CreateLowCutFilter CreateEQ3BandFFT CreateEQ3Band CreateHardDistortion CreateTremolo CreateHighCutFilterGPU "
                  "CreateLowCutFilterGPU CreateEQ3BandFFTGPU").split():
            assert callable(getattr(pyAudioDspTools, n)), n
        compat.install()  # idempotent
    finally:
        compat.uninstall()
    assert "pyAudioDspTools" not in sys.modules and "pyAudioDspTools.Utility" not in sys.modules


def test_compat_refuses_to_shadow_another_package():
    import types
    from pyaudiodsptools_amd import compat
    sys.modules["pyAudioDspTools"] = types.ModuleType("pyAudioDspTools")
    try:
        with pytest.raises(ImportError):
            compat.install()
    finally:
        del sys.modules["pyAudioDspTools"]
