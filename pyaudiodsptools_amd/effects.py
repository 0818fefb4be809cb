"""The reference's elementwise effects on the GPU: VolumeChange, CreateSoftClipper, CreateHardDistortion,
CreateSaturator, CreateTremolo and MixSignals.

Each is one arithmetic expression per sample, so it runs either
  * fused: ``engine.set_epilogue(effect)`` / ``fuse(lowcut, eq, highcut, clipper)`` applies it to the filter kernel's
    output registers before they are stored - no extra pass over HBM; or
  * standalone: ``effect.apply(x)`` launches the elementwise kernel of libadsp (adsp_effect_host / adsp_effect_device).

Same constructor arguments, defaults and ``.apply(array) -> array`` as the reference (EffectSoftClipper.py:18-45,
EffectHardDistortion.py:14-41, EffectSaturator.py:19-49, EffectTremolo.py:19-57, Utility.py:51-72, 171-194).  The arithmetic is float32 (the
reference computes in the dtype it is handed; its FFT devices hand it float32).
"""
import ctypes

import math

import numpy as np

from . import _capi, config


class Effect:
    """op code + three float parameters, exactly what the C ABI takes (include/adsp.h ADSP_EFFECT_*)."""
    op = _capi.EFFECT_NONE

    def params(self):
        return (0.0, 0.0, 0.0)

    def _phase(self, n_samples):
        """LFO table index of the first sample of the next n_samples (effects with a time base only); advances the
        time base.  `_peek_phase` / `_commit_phase` split it so that a failed call leaves the time base where it was."""
        return 0

    def _peek_phase(self, n_samples):
        return 0, None

    def _commit_phase(self, token):
        pass

    def apply(self, float_array_input, device=0, stream=None):
        """numpy array / list -> fresh float32 numpy array of the same shape; a torch CUDA tensor -> fresh CUDA tensor."""
        lib = _capi.load()
        p0, p1, p2 = (float(v) for v in self.params())
        if hasattr(float_array_input, "data_ptr"):  # device-resident (torch is plumbing only)
            x = float_array_input
            if str(x.dtype) != "torch.float32" or not x.is_cuda or not x.is_contiguous():
                raise TypeError("device input must be a contiguous float32 CUDA tensor")
            y = x.new_empty(x.shape)
            phase, token = self._peek_phase(x.numel())
            _capi.check(lib.adsp_effect_device(x.device.index or 0, self.op, p0, p1, p2, phase,
                                               ctypes.c_void_p(x.data_ptr()),
                                               ctypes.c_void_p(y.data_ptr()), x.numel(),
                                               ctypes.c_void_p(stream) if stream else None))
            self._commit_phase(token)  # only a call that went through moves the LFO
            return y
        x = np.ascontiguousarray(float_array_input, dtype=np.float32)
        y = np.empty_like(x)
        phase, token = self._peek_phase(x.size)
        _capi.check(lib.adsp_effect_host(int(device), self.op, p0, p1, p2, phase, ctypes.c_void_p(x.ctypes.data),
                                         ctypes.c_void_p(y.ctypes.data), x.size))
        self._commit_phase(token)
        return y


class CreateSoftClipper(Effect):
    """sign(x) * (1 - |min(|x|, 1) - 1| ** (drive + 1))   (EffectSoftClipper.py:18-45)."""
    op = _capi.EFFECT_SOFT_CLIPPER

    def __init__(self, drive=0.44):
        self.placeholder = True
        self.drive = drive + 1

    def params(self):
        return (self.drive, 0.0, 0.0)


class CreateHardDistortion(Effect):
    """(0.8 + 0.2 sin((a - 0.8) / 0.2)) * sign, a = |x| up to 0.8 and the SIGN beyond it - the reference's formula
    as written, including its asymmetry for loud negative samples (EffectHardDistortion.py:31-41)."""
    op = _capi.EFFECT_HARD_DISTORTION

    def __init__(self):
        self.linear_limit = 0.8


class CreateSaturator(Effect):
    """Knee compression above the threshold, then make-up gain (EffectSaturator.py:19-49)."""
    op = _capi.EFFECT_SATURATOR

    def __init__(self, saturation_threshold_in_db=-20.0, makeup_gain=2.0, mode='hard'):
        self.saturation_coeff = 10 ** (saturation_threshold_in_db / 20)
        self.makeup_gain = makeup_gain
        if mode not in ('soft', 'hard'):
            # the reference leaves self.mode unset here and fails with AttributeError on the first apply()
            raise ValueError("mode must be 'hard' or 'soft'")
        self.mode = 2 if mode == 'soft' else 1

    def params(self):
        return (self.saturation_coeff, 10 ** (self.makeup_gain / 20), float(self.mode))


class CreateBitCrusher(Effect):
    """The reference's private, unexported bit crusher (_EffectBitCrusher.py:3-12): 16-bit quantisation, the lowest 9 bits
    dropped (floor), rescaled by 1/64 - so full scale comes out as +-1.0 in steps of 1/64.  float32 result (the reference
    returns float64; every value is an exact multiple of 1/64 either way)."""
    op = _capi.EFFECT_BIT_CRUSHER

    def __init__(self):
        self.placeholder = True


class _Volume(Effect):
    op = _capi.EFFECT_VOLUME

    def __init__(self, gain_change_in_db, overflow_protection=True):
        self.gain_change_in_db = gain_change_in_db
        self.overflow_protection = overflow_protection

    def params(self):
        return (10 ** (self.gain_change_in_db / 20), 1.0 if self.overflow_protection == True else 0.0, 0.0)  # noqa: E712


def CreateVolumeChange(gain_change_in_db, overflow_protection=True):
    """VolumeChange as a fusable effect object (the reference only has the function form)."""
    return _Volume(gain_change_in_db, overflow_protection)


def VolumeChange(float_array_input, gain_change_in_db, overflow_protection=True):
    """(10 ** (dB / 20)) * x, clipped to [-1, 1] unless overflow_protection is off (Utility.py:171-194)."""
    return _Volume(gain_change_in_db, overflow_protection).apply(float_array_input)


class CreateTremolo(Effect):
    """Amplitude modulation by a periodic LFO table: gain[n] = (sin(2 pi f n / fs) / 2 + 0.5) depth + (1 - depth), the
    table being ``len(arange(float32(fs / f)))`` samples long and repeated end to end (EffectTremolo.py:19-47).

    The only state is where in the table the next sample falls.  It follows the reference's buffer arithmetic
    (:40-45), including its quirk: when the buffer holds exactly one chunk, ``copy[-0:]`` keeps all of it and every
    later chunk replays the same table segment.  Standalone ``apply`` takes 1-D arrays (one stream), like the
    reference.  Fused onto an engine (``engine.set_epilogue(tremolo)``) the table index lives in the engine, starts at
    0 and advances with the chunks the engine filters."""
    op = _capi.EFFECT_TREMOLO

    def __init__(self, tremolo_depth=0.4, lfo_in_hertz=4.5):
        if config.sampling_rate is None:
            raise RuntimeError("call config.initialize(sampling_rate, chunk_size) before creating devices")
        self.sin_sample_rate = config.sampling_rate
        self.tremolo_depth = tremolo_depth
        self.lfo_in_hertz = lfo_in_hertz
        self.lfo_length = int(math.ceil(np.float32(self.sin_sample_rate / lfo_in_hertz)))
        if not 1 <= self.lfo_length <= 1 << 23:
            raise ValueError("LFO period must be between 1 and 2**23 samples")
        self.reset()

    @property
    def sin_lfo(self):
        """The LFO table as the reference exposes it (float32[lfo_length]); the GPU evaluates it in place."""
        n = np.arange(self.lfo_length, dtype=np.float64)
        gain = (np.sin(2 * np.pi * self.lfo_in_hertz * n / self.sin_sample_rate) / 2 + 0.5) * self.tremolo_depth
        return (gain + (1 - self.tremolo_depth)).astype(np.float32)

    def params(self):
        return (self.tremolo_depth, self.lfo_in_hertz / self.sin_sample_rate, float(self.lfo_length))

    def _peek_phase(self, n_samples):
        buffered = self._buffered
        if buffered < n_samples:  # whole tables are appended until the buffer covers the chunk
            buffered += -(-(n_samples - buffered) // self.lfo_length) * self.lfo_length
        phase = (-buffered) % self.lfo_length
        if buffered != n_samples:  # the reference's copy[-0:] keeps everything in that one case
            buffered -= n_samples
        return phase, buffered

    def _commit_phase(self, token):
        self._buffered = token

    def _phase(self, n_samples):
        phase, token = self._peek_phase(n_samples)
        self._commit_phase(token)
        return phase

    def apply(self, float_array_input, device=0, stream=None):
        if len(getattr(float_array_input, "shape", (len(float_array_input),))) != 1:
            raise ValueError("CreateTremolo.apply takes one 1-D stream; fuse it onto an engine for channel batches")
        return super().apply(float_array_input, device=device, stream=stream)

    def reset(self):
        """Restart the LFO (EffectTremolo.py:49-57)."""
        self._buffered = self.lfo_length


def MixSignals(*args, device=0):
    """clip(sum of the signals, -1, 1) (Utility.py:51-72).  float32 result (the reference's is float64)."""
    arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in args]
    if not arrays:
        raise IndexError("tuple index out of range")  # what the reference's args[0] raises
    if any(a.shape != arrays[0].shape for a in arrays):
        raise ValueError("Something went wrong. Make sure, that the Numpy arrays are equal in length.")
    out = np.empty_like(arrays[0])
    ptrs = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
    _capi.check(_capi.load().adsp_mix_host(int(device), ptrs, len(arrays), 1, ctypes.c_void_p(out.ctypes.data), out.size))
    return out
