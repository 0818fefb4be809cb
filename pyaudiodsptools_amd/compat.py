"""`install()` makes this package importable under the reference's name AND module layout, so that a script written against the
reference runs unchanged - its own ModuleTests.py starts with

    import pyAudioDspTools
    pyAudioDspTools.config.initialize(44100, 512)
    from pyAudioDspTools.Generators import CreateSinewave, CreateSquarewave, CreateWhitenoise
    from pyAudioDspTools.Utility import MakeChunks, CombineChunks, MixSignals, ConvertdBVTo16Bit ...
    from pyAudioDspTools.EffectCompressor import CreateCompressor
    from pyAudioDspTools._EffectReverb import CreateReverb            (ModuleTests.py:12, :34-52)

The reference keeps one class per file; here the classes live where their kernels' host code lives (devices.py, effects.py,
recursive.py, delay.py, wavio.py, signals.py).  `install()` registers `pyAudioDspTools` -> this package and one synthetic
module per reference file holding the names that file defines (pyAudioDspTools/__init__.py:11-28).  Nothing is copied: every entry IS the object
of this package.  Call it once, before the first `import pyAudioDspTools`:

    import pyaudiodsptools_amd.compat; pyaudiodsptools_amd.compat.install()
"""
import sys
import types

# reference file -> the names it defines (pyAudioDspTools/__init__.py:11-28; the private files by their own class statements)
LAYOUT = {
    "Generators": ("CreateSinewave", "CreateSquarewave", "CreateWhitenoise"),
    "Utility": ("MakeChunks", "CombineChunks", "MixSignals", "ConvertdBVTo16Bit", "Convert16BitTodBV", "Dither16BitTo8Bit",
                "Dither32BitIntTo16BitInt", "MonoWavToNumpyFloat", "StereoWavToNumpyFloat", "InfodBV", "InfodBV16Bit", "VolumeChange",
                "MonoWavToNumpy16BitInt", "NumpyFloatToWav"),
    "EffectCompressor": ("CreateCompressor",),
    "EffectSoftClipper": ("CreateSoftClipper",),
    "EffectSaturator": ("CreateSaturator",),
    "EffectGate": ("CreateGate",),
    "EffectDelay": ("CreateDelay",),
    "EffectFFTFilter": ("CreateHighCutFilter", "CreateLowCutFilter"),
    "EffectEQ3BandFFT": ("CreateEQ3BandFFT",),
    "EffectEQ3Band": ("CreateEQ3Band",),
    "EffectHardDistortion": ("CreateHardDistortion",),
    "EffectTremolo": ("CreateTremolo",),
    "EffectFFTFilterGPU": ("CreateHighCutFilterGPU", "CreateLowCutFilterGPU"),
    "EffectEQ3BandFFTGPU": ("CreateEQ3BandFFTGPU",),
    "_EffectReverb": ("CreateReverb",),
    "_EffectBitCrusher": ("CreateBitCrusher",),
}


def install(name="pyAudioDspTools"):
    """Register this package and the reference's per-file modules under `name`.  Idempotent; refuses to shadow another package that
    is already imported under that name (the real reference, say)."""
    import pyaudiodsptools_amd as pkg
    have = sys.modules.get(name)
    if have is not None and have is not pkg:
        raise ImportError(f"a different module named {name!r} is already imported ({getattr(have, '__file__', '?')})")
    sys.modules[name] = pkg
    sys.modules[name + ".config"] = pkg.config
    for module_name, names in LAYOUT.items():
        full = f"{name}.{module_name}"
        if module_name == "Utility":  # wavio.py carries every name of the reference's Utility.py (and WavBank)
            mod = pkg.Utility
        else:
            mod = sys.modules.get(full) or types.ModuleType(full, f"{module_name}.py of the reference's layout: names of pyaudiodsptools_amd")
            for n in names:
                setattr(mod, n, getattr(pkg, n))
            mod.config = pkg.config
            setattr(pkg, module_name, mod)
        missing = [n for n in names if not hasattr(mod, n)]
        if missing:
            raise ImportError(f"{full}: {missing} not defined")
        sys.modules[full] = mod
    return pkg


def uninstall(name="pyAudioDspTools"):
    """Remove what install() registered (tests)."""
    import pyaudiodsptools_amd as pkg
    if sys.modules.get(name) is pkg:
        del sys.modules[name]
        for full in [k for k in sys.modules if k.startswith(name + ".")]:
            del sys.modules[full]
