"""The reference's recursive devices on the GPU (last item): CreateEQ3Band (IIR),
CreateCompressor and CreateGate.  Sample-to-sample recurrences cannot be split over time without changing their rounding,
so libadsp runs them as per-channel sequential scans (one lane per channel, adsp_scan_*): the point of these classes is
many channels per call (``channels=``, ``apply_batch``), and chains that stay on the device next to the FFT engines.
"""
import ctypes

import numpy as np

from . import _capi, config
from .engine import _ptr


class ScanEngine:
    """Handle on one adsp_scan engine: [steps, C, N] float32 batches, state carried across calls."""

    def __init__(self, handle, chunk_size, channels, device):
        self._lib = _capi.load()
        self._h = handle
        self.chunk_size, self.channels, self.device = int(chunk_size), int(channels), int(device)

    @classmethod
    def biquad(cls, sections, chunk_size, channels=1, device=0):
        """sections: [n][5] float64 rows b0/a0, b1/a0, b2/a0, a1/a0, a2/a0."""
        co = np.ascontiguousarray(sections, dtype=np.float64).reshape(-1, 5)
        cfg = _capi.AdspScanConfig(int(device), int(chunk_size), int(channels), 0, len(co))
        h = ctypes.c_void_p(None)
        _capi.check(_capi.load().adsp_scan_create_biquad(ctypes.byref(cfg), _ptr(co), ctypes.byref(h)))
        return cls(h, chunk_size, channels, device)

    @classmethod
    def compressor(cls, threshold, attack_envelope, release_envelope, chunk_size, channels=1, device=0):
        att = np.ascontiguousarray(attack_envelope, dtype=np.float32)
        rel = np.ascontiguousarray(release_envelope, dtype=np.float32)
        cfg = _capi.AdspScanConfig(int(device), int(chunk_size), int(channels), 0, 0)
        h = ctypes.c_void_p(None)
        _capi.check(_capi.load().adsp_scan_create_compressor(ctypes.byref(cfg), float(threshold), _ptr(att), len(att), _ptr(rel),
                                                              len(rel), ctypes.byref(h)))
        return cls(h, chunk_size, channels, device)

    @classmethod
    def gate(cls, threshold, depth, attack_envelope, release_envelope, chunk_size, channels=1, device=0):
        """The compressor's state machine on (x * depth), threshold tested on the raw x (EffectGate.py:58-59)."""
        att = np.ascontiguousarray(attack_envelope, dtype=np.float32)
        rel = np.ascontiguousarray(release_envelope, dtype=np.float32)
        cfg = _capi.AdspScanConfig(int(device), int(chunk_size), int(channels), 0, 0)
        h = ctypes.c_void_p(None)
        _capi.check(_capi.load().adsp_scan_create_gate(ctypes.byref(cfg), float(threshold), float(depth), _ptr(att), len(att),
                                                        _ptr(rel), len(rel), ctypes.byref(h)))
        return cls(h, chunk_size, channels, device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.adsp_scan_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _capi.check(self._lib.adsp_scan_reset(self._h))

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """Device-resident [n_steps, C, N] float32 buffers; d_out may be d_in."""
        _capi.check(self._lib.adsp_scan_apply_device(self._h, _ptr(d_in), _ptr(d_out), int(n_steps), _ptr(stream)))

    def apply_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.channels, self.chunk_size):
            raise ValueError(f"expected [steps, {self.channels}, {self.chunk_size}], got {x.shape}")
        y = np.empty_like(x)
        _capi.check(self._lib.adsp_scan_apply_host(self._h, _ptr(x), _ptr(y), x.shape[0]))
        return y[0] if squeeze else y


def _one_chunk(x, n, channels):
    if channels != 1:
        raise ValueError("this device holds several channels; use the *_batch methods with x[channels, chunk]")
    flat = np.concatenate((x,), axis=None)
    if flat.size != n:
        raise ValueError(f"chunk has {flat.size} samples, config.chunk_size was {n} when this device was created")
    return np.ascontiguousarray(flat, dtype=np.float32).reshape(1, n)


class CreateEQ3Band:
    """IIR 3-band EQ after the RBJ cookbook (EffectEQ3Band.py:4-181): low shelf (Q 1), peaking mid (Q 2.5), high shelf
    (Q 1), sampling rate hard-wired to 44100 Hz like the reference, three separate band methods and no ``apply``.

    Each band is the reference's recursion to the bit for float32 chunks: float64 arithmetic, float32 state, and the
    input delayed by one sample (the reference prepends three old inputs but two old outputs).  Results are fresh
    arrays.  ``apply_all_batch`` runs low -> mid -> high in one kernel."""

    def __init__(self, low_shelf_frequency, low_shelf_gain, mid_frequency, mid_gain, high_shelf_frequency,
                 high_shelf_gain, *, channels=1, device=0):
        if config.chunk_size is None:
            raise RuntimeError("call config.initialize(sampling_rate, chunk_size) before creating devices")
        self._n, self.channels = int(config.chunk_size), int(channels)
        self.LowdBgain, self.MiddBgain, self.HighdBgain = low_shelf_gain, mid_gain, high_shelf_gain
        self.Fs = 44100.0
        self.LowShelfFreq, self.MidFreq, self.HighShelfFreq = low_shelf_frequency, mid_frequency, high_shelf_frequency
        self.LowShelfQ, self.MidQ, self.HighShelfQ = 1.0, 2.5, 1.0
        rows = {}
        for name, freq, gain_db, kind in (("LOW", low_shelf_frequency, low_shelf_gain, "lowshelf"),
                                          ("MID", mid_frequency, mid_gain, "peak"),
                                          ("HIGH", high_shelf_frequency, high_shelf_gain, "highshelf")):
            b0, b1, b2, a0, a1, a2 = _rbj(kind, freq, gain_db, self.Fs)
            for key, val in zip(("b0", "b1", "b2", "a0", "a1", "a2"), (b0, b1, b2, a0, a1, a2)):
                setattr(self, name + key, val)  # the reference's attribute names (LOWb0 ... HIGHa2)
            rows[name] = [b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0]
        mk = lambda sec: ScanEngine.biquad(sec, self._n, channels, device)  # noqa: E731
        self._low, self._mid, self._high = mk([rows["LOW"]]), mk([rows["MID"]]), mk([rows["HIGH"]])
        self._all = mk([rows["LOW"], rows["MID"], rows["HIGH"]])

    def applylowband(self, float_array_input):
        return self._low.apply_host(_one_chunk(float_array_input, self._n, self.channels)).reshape(self._n)

    def applymidband(self, float_array_input):
        return self._mid.apply_host(_one_chunk(float_array_input, self._n, self.channels)).reshape(self._n)

    def applyhighband(self, float_array_input):
        return self._high.apply_host(_one_chunk(float_array_input, self._n, self.channels)).reshape(self._n)

    def applylowband_batch(self, x):
        return self._low.apply_host(x)

    def applymidband_batch(self, x):
        return self._mid.apply_host(x)

    def applyhighband_batch(self, x):
        return self._high.apply_host(x)

    def apply_all_batch(self, x):
        """low -> mid -> high shelf in ONE pass (its own filter state, independent of the three band methods)."""
        return self._all.apply_host(x)

    @property
    def cascade(self):
        """The three-section engine behind apply_all_batch, for device-resident chains."""
        return self._all


def _rbj(kind, freq, gain_db, fs):
    a = np.sqrt(10 ** (gain_db / 20))
    w0 = 2 * np.pi * freq / fs
    cw = np.cos(w0)
    if kind == "peak":
        alpha = np.sin(w0) / (2 * 2.5)
        return 1 + alpha * a, -2 * cw, 1 - alpha * a, 1 + alpha / a, -2 * cw, 1 - alpha / a
    alpha = np.sin(w0) / 2 * np.sqrt((a + 1 / a) * (1 / 1.0 - 1) + 2)
    rt = 2 * np.sqrt(a) * alpha
    if kind == "lowshelf":
        return (a * ((a + 1) - (a - 1) * cw + rt), 2 * a * ((a - 1) - (a + 1) * cw), a * ((a + 1) - (a - 1) * cw - rt),
                (a + 1) + (a - 1) * cw + rt, -2 * ((a - 1) + (a + 1) * cw), (a + 1) + (a - 1) * cw - rt)
    return (a * ((a + 1) + (a - 1) * cw + rt), -2 * a * ((a - 1) + (a + 1) * cw), a * ((a + 1) + (a - 1) * cw - rt),
            (a + 1) - (a - 1) * cw + rt, 2 * ((a - 1) - (a + 1) * cw), (a + 1) - (a - 1) * cw - rt)


class CreateCompressor:
    """Drop-in for the reference's compressor (EffectCompressor.py:8-125): same arguments and ``.apply(chunk)``, bit for
    bit including its quirks (no hold time, the sample after a completed release passes untouched, a re-trigger during
    release jumps straight to full compression).  The result is a fresh array; the reference writes into its argument."""

    def __init__(self, threshold_in_db=-15, ratio=0.60, attack_in_ms=3.1, release_in_ms=30.1, *, channels=1, device=0):
        if config.chunk_size is None or config.sampling_rate is None:
            raise RuntimeError("call config.initialize(sampling_rate, chunk_size) before creating devices")
        self._n, self.channels = int(config.chunk_size), int(channels)
        self.ratio = ratio
        self.threshold_power = np.float32(10 ** (threshold_in_db / 20))
        self.attack_envelope = np.linspace(1.0, ratio, num=int((config.sampling_rate / 1000) * attack_in_ms), dtype="float32")
        self.release_envelope = np.linspace(ratio, 1.0, num=int((config.sampling_rate / 1000) * release_in_ms), dtype="float32")
        if len(self.attack_envelope) < 1 or len(self.release_envelope) < 1:
            raise ValueError("attack and release must last at least one sample")
        self.engine = ScanEngine.compressor(self.threshold_power, self.attack_envelope, self.release_envelope, self._n,
                                            channels, device)

    def apply(self, int_array_input):
        return self.engine.apply_host(_one_chunk(int_array_input, self._n, self.channels)).reshape(self._n)

    def apply_batch(self, x):
        return self.engine.apply_host(x)

    def reset(self):
        self.engine.reset()


class CreateGate:
    """Drop-in for the reference's gate (EffectGate.py:6-126): same arguments, defaults and ``.apply(chunk)``.  The
    reference scales a copy of the chunk by ``depth`` (:59), walks the compressor's attack / hold / release loop nest over
    envelopes that run 1 -> 1/depth -> 1 (built for 44100 Hz whatever ``config`` says, :29-33), testing the threshold on
    the unscaled samples (:58), and returns that copy (:126).  Bit for bit the same here for float32 chunks."""

    def __init__(self, threshold_in_db=-5, depth=0.1, attack=3.1, release=200.1, *, channels=1, device=0):
        if config.chunk_size is None:
            raise RuntimeError("call config.initialize(sampling_rate, chunk_size) before creating devices")
        self._n, self.channels = int(config.chunk_size), int(channels)
        self.depth = depth
        self.threshold_power = np.float32(10 ** (threshold_in_db / 20))
        self.attack_envelope = np.linspace(1.0, 1.0 / depth, num=int((44100 / 1000) * attack), dtype="float32")
        self.release_envelope = np.linspace(1.0 / depth, 1.0, num=int((44100 / 1000) * release), dtype="float32")
        if len(self.attack_envelope) < 1 or len(self.release_envelope) < 1:
            raise ValueError("attack and release must last at least one sample")
        self.engine = ScanEngine.gate(self.threshold_power, np.float32(depth), self.attack_envelope, self.release_envelope,
                                      self._n, channels, device)

    def apply(self, int_array_input):
        return self.engine.apply_host(_one_chunk(int_array_input, self._n, self.channels)).reshape(self._n)

    def apply_batch(self, x):
        return self.engine.apply_host(x)

    def reset(self):
        self.engine.reset()
