"""CPU oracle for the stateless wave-shapers.  TEST INFRASTRUCTURE ONLY - see fftfilter_oracle.py.

numpy restatement, in float32 like the reference computes when it is handed float32 chunks (numpy >= 2 keeps python
scalars "weak", so every intermediate stays float32).  Pinned against tests/golden/kat_effects.npz, captured from the
real reference by tests/golden/make_golden.py.

Reference anchors:
  * soft clipper      pyAudioDspTools/EffectSoftClipper.py:18-45
  * hard distortion   pyAudioDspTools/EffectHardDistortion.py:14-41
  * saturator         pyAudioDspTools/EffectSaturator.py:19-49
  * volume change     pyAudioDspTools/Utility.py:171-194
  * tremolo           pyAudioDspTools/EffectTremolo.py:19-57
  * mix               pyAudioDspTools/Utility.py:51-72
"""
import numpy as np

F = np.float32


def soft_clipper(x, drive=0.44):
    x = np.asarray(x, F)
    a = np.minimum(np.abs(x), F(1))
    shaped = F(1) - np.power(np.abs(a - F(1)), F(drive + 1))
    return np.where(x < 0, -shaped, shaped).astype(F)


def hard_distortion(x):
    x = np.asarray(x, F)
    sign = np.where(x >= 0, F(1), F(-1))
    a = np.abs(x)
    a = np.where(a <= F(0.8), a, sign)  # the sign, not 1.0: loud negative samples land on -1 (reference quirk)
    scale = F(1.0 - 0.8)
    return ((F(0.8) + scale * np.sin((a - F(0.8)) / scale)) * sign).astype(F)


def saturator(x, saturation_threshold_in_db=-20.0, makeup_gain=2.0, mode="hard"):
    x = np.asarray(x, F)
    c = F(10 ** (saturation_threshold_in_db / 20))
    power = {"hard": 1, "soft": 2}[mode]
    a = np.abs(x)
    u = a - c
    with np.errstate(all="ignore"):
        knee = c + u / (F(1) + (u / (F(1) - c)) ** power)
    a = np.where(a > c, knee, a)
    a = np.where(a > F(1), (c + F(1)) / F(2), a)
    return (F(10 ** (makeup_gain / 20)) * np.where(x < 0, -a, a)).astype(F)


def bit_crusher(x):
    """_EffectBitCrusher.py:8-12 (private in the reference): 16-bit quantise, drop 9 bits with floor, rescale."""
    q = (np.asarray(x, F) * 32767).astype(np.int16)
    return (q // 512) / 64


def volume_change(x, gain_change_in_db, overflow_protection=True):
    y = F(10 ** (gain_change_in_db / 20)) * np.asarray(x, F)
    return np.clip(y, F(-1), F(1)) if overflow_protection else y


class OracleTremolo:
    """The LFO table repeated end to end, consumed chunk by chunk; `buffered` is the length of the reference's
    sin_lfo_copy, the whole state (the buffer always ends on a table end).  Keeps the reference's quirk: a buffer of
    exactly one chunk is not consumed (``copy[-0:]``), so that chunk's table segment is replayed from then on."""

    def __init__(self, sampling_rate, tremolo_depth=0.4, lfo_in_hertz=4.5):
        n = np.arange(F(sampling_rate / lfo_in_hertz))  # float32 ramp, like the reference's
        self.table = (((np.sin(2 * np.pi * lfo_in_hertz * n / sampling_rate) / 2) + 0.5) * tremolo_depth
                      + (1 - tremolo_depth)).astype(F)
        self.reset()

    def reset(self):
        self.buffered = len(self.table)

    def apply(self, x):
        x = np.asarray(x, F)
        period = len(self.table)
        while self.buffered < len(x):
            self.buffered += period
        idx = ((-self.buffered) % period + np.arange(len(x))) % period
        if self.buffered != len(x):
            self.buffered -= len(x)
        return x * self.table[idx]


def mix_signals(*signals):
    acc = np.zeros(len(signals[0]))  # float64 accumulator, like the reference
    for sig in signals:
        acc = acc + sig
    return np.clip(acc, -1.0, 1.0)
