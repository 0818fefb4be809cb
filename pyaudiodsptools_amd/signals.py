"""Test-signal generators and level / bit-depth helpers of the reference's export surface (Generators.py, Utility.py:74-169) - host-side
numpy, like the chunk and WAV plumbing of wavio.py: they run once, before or after the stream goes through the devices, and are here so
that a script written against the reference (its own ModuleTests.py imports every one of them, :38-41) finds them under the same
names with the same results.  Nothing in this module is on the filter path or in any measured region.

Results are the reference's bit for bit where it is deterministic (pinned by tests/golden/kat_moduletests.npz); its white noise and its
two dither functions draw from numpy's global, unseeded generator - here they take an optional `seed` / `rng` and are held to the
quantities that do not depend on the draw (band edges and magnitude of the noise spectrum; the dither being 0 or -1 LSB).
"""
import math

import numpy as np

from . import config


def _phase(frequency, length_in_samples):
    # the reference's grouping: ((2 pi f) t) / fs in float64, t an integer ramp (Generators.py:25-26)
    return 2 * np.pi * frequency * np.arange(length_in_samples) / config.sampling_rate


def CreateSinewave(sin_frequency, sin_length_in_samples):
    """float32 sine of `sin_frequency` Hz at config.sampling_rate, starting at phase 0 (Generators.py:5-27)."""
    return np.sin(_phase(sin_frequency, sin_length_in_samples)).astype(np.float32)


def CreateSquarewave(square_frequency, square_length_in_samples):
    """+1.0 where the float32 sine is positive, -1.0 elsewhere - zero crossings included; float64 like the reference's
    (Generators.py:30-55)."""
    return np.where(CreateSinewave(square_frequency, square_length_in_samples) > 0, 1.0, -1.0)


def CreateWhitenoise(noise_length_in_samples, seed=None):
    """Noise with a FLAT magnitude spectrum between 20 Hz and 20 kHz and nothing outside: every in-band bin of the n-point
    spectrum has magnitude 1 and a uniformly random phase, the inverse transform is scaled by 5 (Generators.py:58-92) - an rms of
    5 sqrt(2 * bins) / n.  float32.  The reference's phases come from numpy's global generator; `seed` makes the draw repeatable."""
    n = int(noise_length_in_samples)
    freqs = np.fft.rfftfreq(n, 1.0 / config.sampling_rate)
    half = np.where((freqs >= 20) & (freqs <= 20000), 1.0, 0.0).astype(np.complex128)
    n_phases = (n - 1) // 2          # bins 1 .. n_phases carry a phase; DC and (even n) the Nyquist bin stay real
    draw = (np.random.default_rng(seed).random(n_phases) if seed is not None else np.random.rand(n_phases)) * 2 * np.pi
    half[1:n_phases + 1] *= np.cos(draw) + 1j * np.sin(draw)
    return (np.fft.irfft(half, n) * 5).astype(np.float32)


def ConvertdBVTo16Bit(float_array_input):
    """[-1, 1] floats (clipped to that range) -> int16 at 32767 per volt, truncated towards zero (Utility.py:75-78)."""
    return (np.clip(float_array_input, -1.0, 1.0) * (2 ** 15 - 1)).astype(np.int16)


def Convert16BitTodBV(int_array_input):
    """int16 -> float32 volts at 32767 per volt (Utility.py:81-83)."""
    return (np.asarray(int_array_input) / 32767).astype(np.float32)


def _dithered(int_array_input, divisor, limit, rng):
    n = np.size(int_array_input)
    draw = rng.integers(-1, 1, size=n) if rng is not None else np.random.randint(-1, 1, size=n)  # -1 or 0
    return np.clip(np.around(np.asarray(int_array_input) / divisor).astype(np.int64) + draw, -limit, limit)


def Dither16BitTo8Bit(int_array_input, rng=None):
    """round(x / 256) plus a rectangular dither of 0 or -1, clipped to +-127 (Utility.py:86-94).  Like the reference's, the result
    keeps a wide integer dtype (its `astype('int8')` is a discarded expression)."""
    return _dithered(int_array_input, 256, 127, rng)


def Dither32BitIntTo16BitInt(int_array_input, rng=None):
    """round(x / 65535) plus a dither of 0 or -1, clipped to +-32767, int16 (Utility.py:97-106)."""
    return _dithered(int_array_input, 65535, 32767, rng).astype(np.int16)


def _mean_level_db(array, full_scale):
    a = np.asarray(array)
    return 20 * math.log10(np.where(a > 0, a, -a).sum() / a.size / full_scale)


def InfodBV(float_array_input):
    """Mean absolute level in dB re 1.0 (Utility.py:124-146); a silent array raises like math.log10(0)."""
    return _mean_level_db(float_array_input, 1)


def InfodBV16Bit(int_array_input):
    """Mean absolute level in dB re 32767 (Utility.py:148-168).  The reference negates int16 samples in int16, where -32768 stays
    negative; so does this."""
    return _mean_level_db(int_array_input, 32767)
