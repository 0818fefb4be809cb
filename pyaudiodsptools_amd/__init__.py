"""pyaudiodsptools_amd - MI355X-native drop-in for pyAudioDspTools' FFT filter / FFT-EQ devices.

    import pyaudiodsptools_amd as pyAudioDspTools
    pyAudioDspTools.config.initialize(44100, 4096)
    dev = pyAudioDspTools.CreateLowCutFilter(800)
    out_chunk = dev.apply(in_chunk)

The FFT-filter hot path of the reference and its callers run in hand-written HIP kernels behind the C ABI of
include/adsp.h; the reference's remaining exports (test-signal generators, level helpers, chunk / WAV plumbing) are host-side numpy
under the same names, and `pyaudiodsptools_amd.compat.install()` registers the reference's module layout (`pyAudioDspTools.Utility`,
`pyAudioDspTools.EffectFFTFilter`, ...) so that scripts written against it - its ModuleTests.py - run unchanged.  Importing the
package does not need a GPU; creating a device does.
"""
from . import config
from ._capi import AdspError
from .design import FirStream
from .devices import (CreateEQ3BandFFT, CreateEQ3BandFFTGPU, CreateHighCutFilter, CreateHighCutFilterGPU,
                      CreateLowCutFilter, CreateLowCutFilterGPU, fuse)
from .effects import (CreateHardDistortion, CreateSaturator, CreateSoftClipper, CreateTremolo, CreateVolumeChange,
                      Effect, MixSignals, VolumeChange)
from .effects import CreateBitCrusher
from .delay import CreateDelay, CreateReverb, DelayLine
from .signals import (Convert16BitTodBV, ConvertdBVTo16Bit, CreateSinewave, CreateSquarewave, CreateWhitenoise, Dither16BitTo8Bit,
                      Dither32BitIntTo16BitInt, InfodBV, InfodBV16Bit)
from .recursive import CreateCompressor, CreateEQ3Band, CreateGate, ScanEngine
from .engine import ExactFirEngine, FirEngine, MixBus, PartitionedFirEngine, UpolsFirEngine, make_engine
from . import wavio as Utility
from .wavio import (CombineChunks, MakeChunks, MonoWavToNumpy16BitInt, MonoWavToNumpyFloat, NumpyFloatToWav,
                    StereoWavToNumpyFloat, WavBank)

__all__ = ["config", "AdspError", "CreateHighCutFilter", "CreateLowCutFilter", "CreateEQ3BandFFT", "CreateHighCutFilterGPU",
           "CreateLowCutFilterGPU", "CreateEQ3BandFFTGPU", "FirEngine", "ExactFirEngine", "PartitionedFirEngine", "UpolsFirEngine", "make_engine", "FirStream", "fuse", "Utility", "MakeChunks",
           "CombineChunks", "MonoWavToNumpyFloat", "MonoWavToNumpy16BitInt", "StereoWavToNumpyFloat", "NumpyFloatToWav",
           "WavBank", "CreateSoftClipper", "CreateHardDistortion", "CreateSaturator", "VolumeChange", "CreateVolumeChange",
           "Effect", "CreateTremolo", "MixSignals", "MixBus", "CreateDelay", "DelayLine", "CreateEQ3Band", "CreateCompressor", "CreateGate", "ScanEngine",
           "CreateSinewave", "CreateSquarewave", "CreateWhitenoise", "ConvertdBVTo16Bit", "Convert16BitTodBV", "Dither16BitTo8Bit",
           "Dither32BitIntTo16BitInt", "InfodBV", "InfodBV16Bit", "CreateReverb", "CreateBitCrusher"]
__version__ = "0.1.0"
