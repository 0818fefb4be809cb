"""CPU oracle for the FFT filter / FFT 3-band EQ hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch numpy restatement of what the reference's
``CreateHighCutFilter`` / ``CreateLowCutFilter`` / ``CreateEQ3BandFFT`` devices
compute.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``pyaudiodsptools_amd``) never imports anything from ``oracle/``.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here
against golden vectors captured from the real reference imported in the build
container (``tests/golden/make_golden.py``), and against the float64
direct-convolution identity, which is independent of any FFT.

Reference anchors (paths relative to the reference repo root):
  * filter design      pyAudioDspTools/EffectFFTFilter.py:18-47 (high cut), :91-123 (low cut)
  * filter apply       pyAudioDspTools/EffectFFTFilter.py:49-75 / :125-151
  * EQ design          pyAudioDspTools/EffectEQ3BandFFT.py:47-153
  * EQ apply           pyAudioDspTools/EffectEQ3BandFFT.py:156-211
  * chunk plumbing     pyAudioDspTools/Utility.py:8-28 (MakeChunks), :31-48 (CombineChunks)
The FFT itself is third-party: numpy.fft (pocketfft), unpinned in the reference's
setup.py:22; this container has numpy 2.2.6.
"""
from __future__ import annotations

import math

import numpy as np

__all__ = [
    "geometry",
    "windowed_sinc",
    "highcut_taps",
    "lowcut_taps",
    "eq3_band_taps",
    "eq3_composite_taps",
    "padded_spectrum_3n",
    "OracleFFTFilter",
    "OracleHighCut",
    "OracleLowCut",
    "OracleEQ3BandFFT",
    "OracleRfft2N",
    "direct_stream_convolution",
    "make_chunks",
    "combine_chunks",
]


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------
def geometry(chunk_size: int):
    """(L, d, slice_start, slice_end_negative) for a chunk size.

    EffectFFTFilter.py:22-25: ``filter_length = chunk//2 - 1``;
    ``array_slice_value_start = chunk + filter_length//2``;
    ``array_slice_value_end = chunk - filter_length//2`` (used as a negative stop).
    """
    n = int(chunk_size)
    taps = n // 2 - 1
    d = taps // 2
    return taps, d, n + d, n - d


# --------------------------------------------------------------------------------------
# filter design (time-domain taps, float64)
# --------------------------------------------------------------------------------------
def windowed_sinc(cutoff_hz: float, fs: float, taps: int, window: np.ndarray) -> np.ndarray:
    """Unity-DC-gain windowed-sinc low-pass.  EffectFFTFilter.py:28-37."""
    centred = np.arange(taps) - (taps - 1) / 2
    h = np.sinc(2 * cutoff_hz / fs * centred)
    h = h * window
    return h / np.sum(h)


def spectral_inversion(h: np.ndarray) -> np.ndarray:
    """Low-pass -> high-pass by ``delta - h``.  EffectFFTFilter.py:112-113."""
    taps = len(h)
    g = -h
    g[(taps - 1) // 2] += 1
    return g


def highcut_taps(cutoff_hz: float, fs: float, chunk_size: int) -> np.ndarray:
    """EffectFFTFilter.py:18-37: Blackman windowed sinc."""
    taps, _, _, _ = geometry(chunk_size)
    return windowed_sinc(cutoff_hz, fs, taps, np.blackman(taps))


def lowcut_taps(cutoff_hz: float, fs: float, chunk_size: int) -> np.ndarray:
    """EffectFFTFilter.py:91-113: Blackman windowed sinc, spectrally inverted."""
    return spectral_inversion(highcut_taps(cutoff_hz, fs, chunk_size))


def eq3_band_taps(lowshelf_hz, midband_hz, highshelf_hz, fs, chunk_size):
    """The four Kaiser(beta=6) kernels of EffectEQ3BandFFT.py:70-133.

    Returns dict: highshelf (inverted sinc @ 0.75 f, :72-83), lowshelf (sinc @ 1.25 f,
    :95-102), mid_lowpass (sinc @ 1.25 f, :112-119), mid_highpass (inverted sinc @ 0.75 f,
    :122-133).
    """
    taps, _, _, _ = geometry(chunk_size)
    win = np.kaiser(taps, 6.0)
    return {
        "highshelf": spectral_inversion(windowed_sinc(highshelf_hz - highshelf_hz / 4, fs, taps, win)),
        "lowshelf": windowed_sinc(lowshelf_hz + lowshelf_hz / 4, fs, taps, win),
        "mid_lowpass": windowed_sinc(midband_hz + midband_hz / 4, fs, taps, win),
        "mid_highpass": spectral_inversion(windowed_sinc(midband_hz - midband_hz / 4, fs, taps, win)),
    }


def eq3_composite_taps(lowshelf_hz, lowshelf_db, midband_hz, midband_db, highshelf_hz, highshelf_db,
                       fs, chunk_size) -> np.ndarray:
    """Single FIR equivalent of EffectEQ3BandFFT.apply (:179-209), 2L-1 taps.

    ``out = sum_b (g_b - 1) * band_b + dry`` where the mid band is the *product* of two
    spectra (:188) -> time-domain convolution, centred at 2d (kept quirk), and the dry
    middle chunk (:209) is a delta at d.
    """
    taps, d, _, _ = geometry(chunk_size)
    k = eq3_band_taps(lowshelf_hz, midband_hz, highshelf_hz, fs, chunk_size)
    c = np.zeros(2 * taps - 1)
    c[:taps] += (10 ** (highshelf_db / 20) - 1) * k["highshelf"]
    c[:taps] += (10 ** (lowshelf_db / 20) - 1) * k["lowshelf"]
    c += (10 ** (midband_db / 20) - 1) * np.convolve(k["mid_highpass"], k["mid_lowpass"])
    c[d] += 1.0
    return c


def padded_spectrum_3n(taps_td: np.ndarray, chunk_size: int) -> np.ndarray:
    """Zero-pad an L-tap kernel to 3N and FFT it.  EffectFFTFilter.py:45-47.

    The reference appends ``N - L + 1`` zeros (length N+1) and then ``2(N+1) - 3`` more
    (length 3N); only the final length matters.
    """
    n = int(chunk_size)
    buf = np.zeros(3 * n)
    buf[: len(taps_td)] = taps_td
    return np.fft.fft(buf)


# --------------------------------------------------------------------------------------
# stateful devices, literal 3N complex-FFT form (what the reference does per call)
# --------------------------------------------------------------------------------------
class OracleFFTFilter:
    """One mono device.  apply() follows EffectFFTFilter.py:63-75 step by step."""

    def __init__(self, taps_td: np.ndarray, chunk_size: int):
        self.n = int(chunk_size)
        _, _, self.start, self.end = geometry(self.n)
        self.spectrum = padded_spectrum_3n(taps_td, self.n)
        # EffectFFTFilter.py:40-42: three float64 zero chunks.
        self.newest = np.zeros(self.n)
        self.middle = np.zeros(self.n)
        self.oldest = np.zeros(self.n)

    def apply(self, chunk):
        # :63-65 rotate; the reference keeps references, so dtype promotion follows the caller's arrays.
        self.oldest, self.middle, self.newest = self.middle, self.newest, chunk
        # :67-68 concatenate (axis=None flattens anything array-like)
        joined = np.concatenate((self.oldest, self.middle, self.newest), axis=None)
        # :70-72
        y = np.fft.ifft(np.fft.fft(joined) * self.spectrum)
        # :73,75
        return y[self.start:-self.end].real.astype("float32")


class OracleHighCut(OracleFFTFilter):
    def __init__(self, cutoff_hz=8000, fs=44100, chunk_size=4096):
        super().__init__(highcut_taps(cutoff_hz, fs, chunk_size), chunk_size)


class OracleLowCut(OracleFFTFilter):
    def __init__(self, cutoff_hz=160, fs=44100, chunk_size=4096):
        super().__init__(lowcut_taps(cutoff_hz, fs, chunk_size), chunk_size)


class OracleEQ3BandFFT:
    """apply() follows EffectEQ3BandFFT.py:171-211: 1 fft, 3 products, 3 iffts, gains, + dry."""

    def __init__(self, lowshelf_hz, lowshelf_db, midband_hz, midband_db, highshelf_hz, highshelf_db,
                 fs=44100, chunk_size=512):
        self.n = int(chunk_size)
        _, _, self.start, self.end = geometry(self.n)
        k = eq3_band_taps(lowshelf_hz, midband_hz, highshelf_hz, fs, self.n)
        self.spec = {name: padded_spectrum_3n(h, self.n) for name, h in k.items()}
        self.g_low = 10 ** (lowshelf_db / 20)
        self.g_mid = 10 ** (midband_db / 20)
        self.g_high = 10 ** (highshelf_db / 20)
        self.newest = np.zeros(self.n)
        self.middle = np.zeros(self.n)
        self.oldest = np.zeros(self.n)

    def _band(self, spectrum_product, gain):
        y = np.fft.ifft(spectrum_product)[self.start:-self.end]
        return (y * gain) - y  # :195,200,205

    def apply(self, chunk):
        self.oldest, self.middle, self.newest = self.middle, self.newest, chunk
        x = np.fft.fft(np.concatenate((self.oldest, self.middle, self.newest), axis=None))  # :175-179
        high = self._band(x * self.spec["highshelf"], self.g_high)                       # :182,193-195
        low = self._band(x * self.spec["lowshelf"], self.g_low)                          # :185,198-200
        mid = self._band(x * (self.spec["mid_highpass"] * self.spec["mid_lowpass"]), self.g_mid)  # :188,203-205
        out = mid + self.middle + low + high                                             # :209
        return out.real.astype("float32")                                                # :211


# --------------------------------------------------------------------------------------
# equivalent formulations (used to cross-check the oracle and as the honest CPU baseline)
# --------------------------------------------------------------------------------------
def direct_stream_convolution(taps_td: np.ndarray, stream: np.ndarray, chunk_size: int,
                              latency_chunks: int = 1, lookahead: int | None = None) -> np.ndarray:
    """float64 ground truth, no FFT: ``out[tau] = sum_t c[t] * s[tau - latency*N + lookahead - t]``.

    ``stream`` is the concatenation of all input chunks (zero history before it); returns the
    concatenation of all output chunks (same length).  For one device latency=1, lookahead=d
    (SURVEY section 0.1/0.2).
    """
    n = int(chunk_size)
    if lookahead is None:
        lookahead = geometry(n)[1]
    s = np.asarray(stream, dtype=np.float64)
    full = np.convolve(s, np.asarray(taps_td, dtype=np.float64))  # full[p] = sum_t c[t] s[p - t]
    shift = latency_chunks * n - lookahead
    out = np.zeros(len(s))
    # out[tau] = full[tau - shift] for tau - shift >= 0
    if shift < len(s):
        out[shift:] = full[: len(s) - shift]
    return out


class OracleRfft2N:
    """Overlap-save with a 2N-point real FFT: the arithmetic the HIP kernel performs, on the CPU.

    Not the reference's literal form; equal to it up to float rounding (tests assert this).
    Supports a batch of channels: apply(x[C, N]) -> y[C, N] float32.
    """

    def __init__(self, taps_td: np.ndarray, chunk_size: int, channels: int = 1, lookahead: int | None = None,
                 dtype=np.float32):
        self.n = n = int(chunk_size)
        d = geometry(n)[1] if lookahead is None else lookahead
        self.f = f = 2 * n
        m = len(taps_td)
        if m > n + 1:
            raise ValueError("kernel too long for a 2N transform")
        # In the reference's 3N buffer (chunks k-2, k-1, k) output i sits at position N+d+i.
        # A 2N window starting at ws sees it at circular index j = N+d+i-ws, which is free of
        # wrap-around when j >= m-1.
        self.ws = max(0, n + d - (m - 1))
        self.j0 = n + d - self.ws
        assert self.ws + f <= 3 * n and self.j0 + n <= f
        self.cdtype = np.complex64 if dtype == np.float32 else np.complex128
        self.spectrum = np.fft.rfft(np.asarray(taps_td, dtype=np.float64), f).astype(self.cdtype)
        self.hist = np.zeros((channels, 3 * n), dtype=dtype)

    def apply(self, chunk_batch):
        x = np.asarray(chunk_batch, dtype=self.hist.dtype).reshape(self.hist.shape[0], self.n)
        self.hist = np.concatenate((self.hist[:, self.n:], x), axis=1)
        win = self.hist[:, self.ws:self.ws + self.f]
        y = np.fft.irfft(np.fft.rfft(win, axis=1) * self.spectrum, self.f, axis=1)
        return y[:, self.j0:self.j0 + self.n].astype(np.float32)


# --------------------------------------------------------------------------------------
# chunk plumbing (Example1 harness)
# --------------------------------------------------------------------------------------
def make_chunks(signal: np.ndarray, chunk_size: int):
    """Utility.py:22-28, including its quirk: the pad test is ``len % number_of_chunks``."""
    number_of_chunks = math.ceil(np.float32(len(signal) / chunk_size))
    if len(signal) % number_of_chunks != 0:
        pad = chunk_size - (len(signal) % chunk_size)
        signal = np.append(signal, np.zeros(pad, dtype="float32"))
    return np.split(signal, number_of_chunks)


def combine_chunks(chunks):
    """Utility.py:45-48 (without the O(n^2) append)."""
    return np.concatenate([np.asarray(c, dtype="float32") for c in chunks]) if len(chunks) else np.array([], "float32")


# --------------------------------------------------------------------------------------
# 16-bit PCM front end: what Example1.py / Example2.py do around the devices
# --------------------------------------------------------------------------------------
def pcm16_to_float(pcm):
    """Utility.py:236-237: int16 -> float32 / 32768."""
    return np.asarray(pcm, dtype=np.int16).astype("float32") / 32768


def float_to_pcm16(x):
    """Utility.py:306: (x * 32767).astype('int16') (truncation toward zero)."""
    return (np.asarray(x) * 32767).astype("int16")


def run_device_pcm16(device, pcm, chunk_size):
    """int16 stream in, int16 stream out through a float device, chunk by chunk (Example1.py:6-22)."""
    x = pcm16_to_float(pcm)
    y = np.concatenate([device.apply(x[i * chunk_size:(i + 1) * chunk_size]) for i in range(len(x) // chunk_size)])
    return float_to_pcm16(y)
