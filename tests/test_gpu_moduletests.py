"""The reference's own harness, ModuleTests.py, against this package on the GPU: the same import lines (through
pyaudiodsptools_amd.compat.install, which registers the reference's module layout), the same devices with the same arguments
(ModuleTests.py:73-84) and the same ten loops over ONE list of chunks (:95-214), compared with what the reference computes in it
(tests/golden/kat_moduletests.npz, made by tests/golden/make_golden_moduletests.py; tests/test_moduletests_oracle.py is the CPU twin that
pins the oracle on the same file)."""
import sys

import numpy as np
import pytest

from conftest import assert_parity, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref_layout():
    import pyaudiodsptools_amd.compat as compat
    from pyaudiodsptools_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback by design)"
    compat.install()
    import pyAudioDspTools
    keep = (pyAudioDspTools.config.sampling_rate, pyAudioDspTools.config.chunk_size)
    pyAudioDspTools.config.initialize(44100, 512)                                      # ModuleTests.py:34
    yield pyAudioDspTools
    pyAudioDspTools.config.sampling_rate, pyAudioDspTools.config.chunk_size = keep
    compat.uninstall()


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat_moduletests")


def harness_devices():
    """ModuleTests.py:38-52 (imports) and :73-84 (one device of every kind), in the order the loops use them."""
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
    harddistortiontest = CreateHardDistortion()
    tremolotest = CreateTremolo()
    delaytest = CreateDelay()
    compressortest = CreateCompressor()
    softclippertest = CreateSoftClipper()
    saturatortest = CreateSaturator()
    gatetest = CreateGate()
    CreateReverb()                                      # created at :80, never applied
    lowcuttest = CreateLowCutFilter(200)
    highcuttest = CreateHighCutFilter(8000)
    CreateEQ3Band(100, 2, 700, -4, 8000, 5)             # likewise (:83)
    eq3bandffttest = CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5)
    return [saturatortest, compressortest, delaytest, tremolotest, harddistortiontest, gatetest, lowcuttest, highcuttest,
            eq3bandffttest, softclippertest]


# bit for bit on the GPU: the state machines and the float32 multiply-adds of the delay line / LFO table; the others evaluate pow / sin /
# an FFT in float32 and are held to the suite's criterion (north_star: 1e-5)
EXACT = {"Compressor", "Delay", "Gate"}


def test_generators_and_helpers_under_the_references_names(ref_layout, kat):
    from pyAudioDspTools.Generators import CreateSinewave, CreateSquarewave, CreateWhitenoise
    from pyAudioDspTools.Utility import MakeChunks, CombineChunks, MixSignals, ConvertdBVTo16Bit
    from pyAudioDspTools.Utility import Convert16BitTodBV, Dither16BitTo8Bit, Dither32BitIntTo16BitInt, MonoWavToNumpyFloat, InfodBV
    from pyAudioDspTools.Utility import InfodBV16Bit, VolumeChange, MonoWavToNumpy16BitInt, NumpyFloatToWav
    n = int(kat["length"])
    sine_full = CreateSinewave(1000, n)                                                 # ModuleTests.py:57-61
    square_full = CreateSquarewave(1000, n)
    noise_full = CreateWhitenoise(n)
    assert np.array_equal(sine_full, kat["sine"]) and sine_full.dtype == np.float32
    assert np.array_equal(square_full.astype(np.int8), kat["square"]) and str(square_full.dtype) == str(kat["square_dtype"])
    assert str(noise_full.dtype) == str(kat["noise_dtype"]) and len(noise_full) == n
    assert np.array_equal(np.concatenate(MakeChunks(sine_full.copy())), kat["stage_00"])
    assert np.array_equal(CombineChunks(MakeChunks(sine_full.copy())), kat["stage_00"])
    assert np.array_equal(ConvertdBVTo16Bit(sine_full * 1.5), kat["to16"])
    assert np.array_equal(Convert16BitTodBV(kat["to16"]), kat["from16"])
    assert InfodBV(sine_full) == float(kat["info_dbv"]) and InfodBV16Bit(kat["to16"]) == float(kat["info_db16"])
    vol = VolumeChange(sine_full, 3.0)                                                 # the GPU's elementwise kernel
    assert vol.dtype == np.float32
    assert_parity(vol, kat["volume_p3db"], what="VolumeChange(+3 dB)")
    assert_parity(MixSignals(sine_full, sine_full), np.clip(2.0 * sine_full.astype(np.float64), -1, 1), what="MixSignals")
    assert all(callable(f) for f in (Dither16BitTo8Bit, Dither32BitIntTo16BitInt, MonoWavToNumpyFloat, MonoWavToNumpy16BitInt, NumpyFloatToWav))


def test_each_device_on_the_references_own_input(ref_layout, kat):
    """Stage k of the harness in isolation: device k of this package on what the REFERENCE handed its device k."""
    names = [str(s) for s in kat["stage_names"]]
    for k, dev in enumerate(harness_devices(), start=1):
        chunks = np.split(kat[f"stage_{k - 1:02d}"].copy(), len(kat["stage_00"]) // 512)
        for counter in range(len(chunks)):
            chunks[counter] = dev.apply(chunks[counter])
        got, want = np.concatenate(chunks), kat[f"stage_{k:02d}"]
        assert got.dtype == np.float32 and got.shape == want.shape, names[k - 1]
        if names[k - 1] in EXACT:
            assert np.array_equal(got, want), (names[k - 1], int(np.argmax(got != want)))
        else:
            assert_parity(got, want, what=names[k - 1])


def test_the_whole_harness_end_to_end(ref_layout, kat):
    """ModuleTests.py:64-217 as written: a copy of the sine is chunked once and the SAME list goes through all ten loops, every entry
    overwritten with the device's output; the combined result is the reference's."""
    import copy
    from pyAudioDspTools.Generators import CreateSinewave
    from pyAudioDspTools.Utility import MakeChunks, CombineChunks
    sine_full = CreateSinewave(1000, int(kat["length"]))
    sine_copy = copy.deepcopy(sine_full)
    sine_chunked = MakeChunks(sine_copy)
    names = [str(s) for s in kat["stage_names"]]
    for k, dev in enumerate(harness_devices(), start=1):
        counter = 0
        for counter in range(len(sine_chunked)):
            sine_chunked[counter] = dev.apply(sine_chunked[counter])
            counter += 1
        # rounding differences of one stage are inputs of the next: the chain is held to the same criterion at every stage
        assert_parity(np.concatenate(sine_chunked), kat[f"stage_{k:02d}"], what=f"after {names[k - 1]}")
    sine_copy = CombineChunks(sine_chunked)
    assert sine_copy.dtype == np.float32 and len(sine_copy) == len(kat["stage_10"]) and np.abs(sine_copy).max() <= 1.0
    assert np.array_equal(sine_full, kat["sine"]), "the generator's array is not touched by the loops"
    assert "pyAudioDspTools" in sys.modules and sys.modules["pyAudioDspTools"].__name__ == "pyaudiodsptools_amd"


@pytest.mark.parametrize("alias_history", [False, True])
def test_the_gpu_harness(ref_layout, kat, alias_history):
    """ModuleTestsGPU.py as written, torch CUDA tensors standing in for cupy arrays: chunk 88200 (:35), the chunked sine as ONE 2-D device
    array (:58), LowCutGPU(200) -> HighCutGPU(8000) -> EQ3BandFFTGPU (:63-65), every loop `arr[i] = dev.apply(arr[i])` (:78-110).  The
    reference's devices keep views of the rows, so its loops filter their own previous outputs: the script's 1 kHz sine, which all three
    devices pass, leaves it 220 dB down (kat: gpu_inplace_3).  Default here = the filtered stream (the documented divergence,
    INTEGRATION.md section 1); alias_history=True = what the script computes."""
    import copy
    import torch
    pyAudioDspTools = ref_layout
    chunks, n, dec = (int(v) for v in kat["gpu_shape"])
    pyAudioDspTools.config.initialize(44100, n, use_gpu=True)                           # ModuleTestsGPU.py:35
    try:
        from pyAudioDspTools import config
        from pyAudioDspTools.Generators import CreateSinewave
        from pyAudioDspTools.Utility import MakeChunks
        from pyAudioDspTools.EffectFFTFilterGPU import CreateHighCutFilterGPU, CreateLowCutFilterGPU
        from pyAudioDspTools.EffectEQ3BandFFTGPU import CreateEQ3BandFFTGPU
        assert config.use_gpu is True and config._gpu_available is True and config.chunk_size == n
        sine_full = CreateSinewave(1000, chunks * n)
        sine_copy = copy.deepcopy(sine_full)
        sine_chunked = torch.from_numpy(np.array(MakeChunks(sine_copy))).cuda()        # cupy.array(MakeChunks(sine_copy)) in the script
        extra = {"alias_history": True} if alias_history else {}
        lowcuttest = CreateLowCutFilterGPU(200, **extra)
        highcuttest = CreateHighCutFilterGPU(8000, **extra)
        eq3bandffttestgpu = CreateEQ3BandFFTGPU(100, 2, 700, -4, 8000, 5, **extra)
        before = 1.0                                                                    # magnitude of what the loop's windows hold
        for k, dev in enumerate((lowcuttest, highcuttest, eq3bandffttestgpu), start=1):
            counter = 0
            for counter in range(len(sine_chunked)):
                sine_chunked[counter] = dev.apply(sine_chunked[counter])
                counter += 1
            got = sine_chunked.cpu().numpy().reshape(-1)[::dec]
            if not alias_history:
                assert_parity(got, kat[f"gpu_clean_{k}"], what=f"loop {k}: the filtered stream")
                continue
            # a feedback loop (each call re-filters earlier outputs) fed by the previous loop's rounding: north_star's bound on the
            # magnitude of the data in the windows - the rows as the loop found them - for this loop's rounding plus the previous loop's
            want = kat[f"gpu_inplace_{k}"]
            err = np.abs(got.astype(np.float64) - want).max()
            assert err <= 2e-5 * before, (k, err, before)
            before = float(np.abs(want).max())
        if alias_history:
            assert np.abs(got).max() < 1e-6      # the script's result: silence
        else:
            assert np.abs(got).max() > 1.0       # the filtered sine (the EQ's +2 ... +5 dB shelves around a 1 kHz tone)
    finally:
        pyAudioDspTools.config.initialize(44100, 512)
