"""Stateless wave-shapers of the reference as GPU effects: VolumeChange, CreateSoftClipper, CreateHardDistortion,
CreateSaturator.

Each is one arithmetic expression per sample, so it runs either
  * fused: ``engine.set_epilogue(effect)`` / ``fuse(lowcut, eq, highcut, clipper)`` applies it to the filter kernel's
    output registers before they are stored - no extra pass over HBM; or
  * standalone: ``effect.apply(x)`` launches the elementwise kernel of libadsp (adsp_effect_host / adsp_effect_device).

Same constructor arguments, defaults and ``.apply(array) -> array`` as the reference (EffectSoftClipper.py:18-45,
EffectHardDistortion.py:14-41, EffectSaturator.py:19-49, Utility.py:171-194).  The arithmetic is float32 (the
reference computes in the dtype it is handed; its FFT devices hand it float32).
"""
import ctypes

import numpy as np

from . import _capi


class Effect:
    """op code + three float parameters, exactly what the C ABI takes (include/adsp.h ADSP_EFFECT_*)."""
    op = _capi.EFFECT_NONE

    def params(self):
        return (0.0, 0.0, 0.0)

    def apply(self, float_array_input, device=0, stream=None):
        """numpy array / list -> fresh float32 numpy array of the same shape; a torch CUDA tensor -> fresh CUDA tensor."""
        lib = _capi.load()
        p0, p1, p2 = (float(v) for v in self.params())
        if hasattr(float_array_input, "data_ptr"):  # device-resident (torch is plumbing only)
            x = float_array_input
            if str(x.dtype) != "torch.float32" or not x.is_cuda or not x.is_contiguous():
                raise TypeError("device input must be a contiguous float32 CUDA tensor")
            y = x.new_empty(x.shape)
            _capi.check(lib.adsp_effect_device(x.device.index or 0, self.op, p0, p1, p2, ctypes.c_void_p(x.data_ptr()),
                                               ctypes.c_void_p(y.data_ptr()), x.numel(),
                                               ctypes.c_void_p(stream) if stream else None))
            return y
        x = np.ascontiguousarray(float_array_input, dtype=np.float32)
        y = np.empty_like(x)
        _capi.check(lib.adsp_effect_host(int(device), self.op, p0, p1, p2, ctypes.c_void_p(x.ctypes.data),
                                         ctypes.c_void_p(y.ctypes.data), x.size))
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
