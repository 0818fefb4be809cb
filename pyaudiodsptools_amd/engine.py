"""FirEngine: Python handle on one libadsp engine (C channels of one streaming FIR on one GPU).

This is the batched form of the reference's ``.apply()``: where the reference runs one Python
object per mono channel and one numpy FFT pipeline per chunk (EffectFFTFilter.py:49-75), an engine
holds the history of C channels in HBM and filters a whole ``[steps, C, N]`` batch per launch.
"""
import ctypes

import numpy as np

from . import _capi
from .design import PCM16_GAIN, FirStream, engine_spectrum, overlap_save_geometry

_FORMATS = {"f32": (_capi.ADSP_FORMAT_F32, np.float32), "s16": (_capi.ADSP_FORMAT_S16, np.int16)}


def _ptr(x):
    """Device/host address of a torch tensor, numpy array, or a raw integer address."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if hasattr(x, "data_ptr"):  # torch tensor (plumbing only)
        return ctypes.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        return ctypes.c_void_p(x.ctypes.data)
    raise TypeError(f"cannot take the address of {type(x)}")


class FirEngine:
    def __init__(self, fir: FirStream, channels=1, device=0, ring_slots=0, fft_mult=0, sample_format="f32",
                 optimize_for="stream"):
        self._lib = _capi.load()
        self._h = ctypes.c_void_p(None)
        self.fir = fir
        self.fft_mult = int(fft_mult)
        if sample_format not in _FORMATS:
            raise ValueError("sample_format must be 'f32' or 's16'")
        self.sample_format = sample_format
        self._fmt_code, self.dtype = _FORMATS[sample_format]
        # int16 engines: (float)x in, (int16)trunc(y) out; the reference's /32768 and *32767 live in the spectrum
        self.gain = PCM16_GAIN if sample_format == "s16" else 1.0
        self.optimize_for = optimize_for
        self.geometry = geo = overlap_save_geometry(fir, self.fft_mult, optimize_for)
        self.chunk_size = int(fir.chunk_size)
        self.channels = int(channels)
        self.device = int(device)
        cfg = _capi.AdspConfig(self.device, self.chunk_size, self.channels, geo.fft_size, geo.history_chunks,
                               geo.lookback, geo.out_offset, int(ring_slots), self._fmt_code)
        _capi.check(self._lib.adsp_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.ring_slots = int(ring_slots) if ring_slots else max(2 * geo.history_chunks, geo.history_chunks + 1)
        self.plan = _capi.plan_describe(self.chunk_size, geo.fft_size)
        self.set_fir(fir)
        self.block_outputs = self.chunk_size
        self.set_block_outputs(geo.max_block_outputs)

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.adsp_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- filter -------------------------------------------------------------------------------
    def set_fir(self, fir: FirStream):
        """Change the filter without touching the history (same geometry required)."""
        geo = overlap_save_geometry(fir, self.fft_mult, self.optimize_for)
        if geo != self.geometry:
            raise ValueError("new filter needs a different transform geometry; create a new engine")
        self.fir = fir
        self.spectrum = engine_spectrum(fir, geo, self.gain)
        self.upload_spectrum(self.spectrum)

    def upload_spectrum(self, spectrum_f32):
        spec = np.ascontiguousarray(spectrum_f32, dtype=np.float32)
        _capi.check(self._lib.adsp_set_spectrum(self._h, _ptr(spec), spec.size // 2))

    def upload_spectrum_device(self, d_spectrum, n_bins, stream=None):
        _capi.check(self._lib.adsp_set_spectrum_device(self._h, _ptr(d_spectrum), int(n_bins), _ptr(stream)))

    def set_block_outputs(self, v):
        _capi.check(self._lib.adsp_set_block_outputs(self._h, int(v)))
        self.block_outputs = int(v)

    # -- state --------------------------------------------------------------------------------
    def reset(self):
        _capi.check(self._lib.adsp_reset(self._h))

    def get_state(self):
        out = np.empty((self.geometry.history_chunks, self.channels, self.chunk_size), self.dtype)
        _capi.check(self._lib.adsp_get_state(self._h, _ptr(out)))
        return out

    def set_state(self, history):
        h = np.ascontiguousarray(history, dtype=self.dtype)
        if h.shape != (self.geometry.history_chunks, self.channels, self.chunk_size):
            raise ValueError(f"state must have shape {(self.geometry.history_chunks, self.channels, self.chunk_size)}")
        _capi.check(self._lib.adsp_set_state(self._h, _ptr(h)))

    # -- apply --------------------------------------------------------------------------------
    def apply_host(self, x):
        """x: host array [steps, C, N] (or [C, N]) of the engine's sample type -> same shape, fresh array."""
        if self.sample_format == "s16" and np.asarray(x).dtype != np.int16:
            raise TypeError("this engine filters int16 PCM; pass an int16 array")
        x = np.ascontiguousarray(x, dtype=self.dtype)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.channels, self.chunk_size):
            raise ValueError(f"expected [steps, {self.channels}, {self.chunk_size}], got {x.shape}")
        out = np.empty_like(x)
        _capi.check(self._lib.adsp_apply_host(self._h, _ptr(x), _ptr(out), x.shape[0]))
        return out[0] if squeeze else out

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """Asynchronous, device-resident [n_steps, C, N] float32 buffers (torch tensors or addresses)."""
        _capi.check(self._lib.adsp_apply_device(self._h, _ptr(d_in), _ptr(d_out), int(n_steps), _ptr(stream)))

    def ring_acquire(self):
        """Device address of the ring slot the producer must fill with the next [C, N] batch."""
        p = ctypes.c_void_p(None)
        _capi.check(self._lib.adsp_ring_acquire(self._h, ctypes.byref(p)))
        return p.value

    def apply_ring(self, d_out, stream=None):
        _capi.check(self._lib.adsp_apply_ring(self._h, _ptr(d_out), _ptr(stream)))

    def enable_kernel_timing(self, enable=True):
        _capi.check(self._lib.adsp_enable_kernel_timing(self._h, 1 if enable else 0))

    def kernel_time(self):
        """(total kernel milliseconds, launches) since the last call; kernel only, HIP events on the launch stream."""
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        _capi.check(self._lib.adsp_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def synchronize(self, stream=None):
        _capi.check(self._lib.adsp_synchronize(self._h, _ptr(stream)))
