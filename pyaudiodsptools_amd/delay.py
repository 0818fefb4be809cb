"""Callers that embed the FFT filters: CreateDelay and the reverb's delay lines.

The reference's delay (EffectDelay.py:31-74) adds each chunk, scaled by ``linspace(0.5, 0.1, loops)[k]``, into a buffer
``T*(k+1)`` samples ahead and returns the head of the buffer plus the chunk: a sparse FIR.  ``DelayLine`` is the GPU form
(libadsp's tapped delay line: input history ring in HBM, one gather kernel), with the same ``[steps, channels, chunk]``
float32 batches as the FFT engines so filter -> delay chains stay on the device.
"""
import ctypes

import numpy as np

from . import _capi, config
from .devices import CreateHighCutFilter, CreateLowCutFilter
from .engine import _ptr


class DelayLine:
    """out[t] = dry * x[t] + sum_k gains[k] * x[t - delays[k]] for `channels` independent channels on one GPU."""

    def __init__(self, delays, gains, dry=1.0, chunk_size=None, channels=1, device=0):
        self._lib = _capi.load()
        self._h = ctypes.c_void_p(None)
        self.chunk_size = int(config.chunk_size if chunk_size is None else chunk_size)
        self.channels, self.device = int(channels), int(device)
        self.delays = np.ascontiguousarray(delays, dtype=np.int32).reshape(-1)
        self.gains = np.ascontiguousarray(gains, dtype=np.float32).reshape(-1)
        if self.delays.shape != self.gains.shape:
            raise ValueError("one gain per tap")
        self.dry = float(dry)
        cfg = _capi.AdspDelayConfig(self.device, self.chunk_size, self.channels, len(self.delays))
        _capi.check(self._lib.adsp_delay_create(ctypes.byref(cfg), _ptr(self.delays) if len(self.delays) else None,
                                                _ptr(self.gains) if len(self.gains) else None, self.dry, ctypes.byref(self._h)))
        h = ctypes.c_int(0)
        _capi.check(self._lib.adsp_delay_history_chunks(self._h, ctypes.byref(h)))
        self.history_chunks = h.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.adsp_delay_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _capi.check(self._lib.adsp_delay_reset(self._h))

    def set_accumulate(self, on=True):
        _capi.check(self._lib.adsp_delay_set_accumulate(self._h, 1 if on else 0))

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """Device-resident [n_steps, C, N] float32 buffers (torch tensors or addresses), asynchronous; not in place."""
        _capi.check(self._lib.adsp_delay_apply_device(self._h, _ptr(d_in), _ptr(d_out), int(n_steps), _ptr(stream)))

    def apply_host(self, x, out=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.channels, self.chunk_size):
            raise ValueError(f"expected [steps, {self.channels}, {self.chunk_size}], got {x.shape}")
        y = np.empty_like(x) if out is None else out.reshape(x.shape)
        _capi.check(self._lib.adsp_delay_apply_host(self._h, _ptr(x), _ptr(y), x.shape[0]))
        return y[0] if squeeze else y


class CreateDelay:
    """Drop-in for the reference's CreateDelay (EffectDelay.py:6-74): same arguments, defaults, attributes and
    ``.apply(chunk)``; ``channels=`` / ``device=`` make it a bank for ``apply_batch``.

    Differences, all deliberate: the result is a fresh array (the reference adds into the caller's chunk in place,
    :66-67); any chunk length the object was created for works (the reference's buffer is too short when the chunk is
    longer than two delay times); and ``use_lowcut_filter`` / ``use_highcut_filter`` WORK - the reference calls
    ``applylowcutfilter`` / ``applyhighcutfilter``, which do not exist (:56,58 -> AttributeError); here they run the
    filters' ``apply`` first, as the reference's own reverb delay line does (_EffectReverb.py:41-44)."""

    def __init__(self, time_in_ms=500, feedback_loops=2, lowcut_filter_frequency=40, highcut_filter_frequency=12000,
                 use_lowcut_filter=False, use_highcut_filter=False, wet=False, *, channels=1, device=0):
        if config.chunk_size is None or config.sampling_rate is None:
            raise RuntimeError("call config.initialize(sampling_rate, chunk_size) before creating devices")
        self._n = int(config.chunk_size)
        self.channels = int(channels)
        self.time_in_samples = int(time_in_ms * (config.sampling_rate / 1000))
        self.wet = wet
        self.max_samples = self.time_in_samples * (feedback_loops + 2)
        self.feedback_ramp = np.linspace(0.5, 0.1, num=feedback_loops, dtype="float32")
        self.use_lowcut_filter = use_lowcut_filter
        self.use_highcut_filter = use_highcut_filter
        # the reference builds both filters whether or not it uses them; only the used ones cost GPU memory here
        self.LowCutFilter = CreateLowCutFilter(lowcut_filter_frequency, channels=channels, device=device) if use_lowcut_filter else None
        self.HighcutFilter = CreateHighCutFilter(highcut_filter_frequency, channels=channels, device=device) if use_highcut_filter else None
        if self.time_in_samples < 1 and feedback_loops > 0:
            raise ValueError("the delay time must be at least one sample")
        delays = [self.time_in_samples * (k + 1) for k in range(len(self.feedback_ramp))]
        self.line = DelayLine(delays, self.feedback_ramp, dry=0.0 if wet else 1.0, chunk_size=self._n, channels=channels,
                              device=device)

    def _filters(self):
        return [f for f in (self.LowCutFilter, self.HighcutFilter) if f is not None]

    def apply(self, float32_array_input):
        if self.channels != 1:
            raise ValueError("this device holds several channels; use apply_batch(x[channels, chunk])")
        flat = np.concatenate((float32_array_input,), axis=None)
        if flat.size != self._n:
            raise ValueError(f"chunk has {flat.size} samples, config.chunk_size was {self._n} when this device was created")
        return self.apply_batch(flat.reshape(1, self._n)).reshape(self._n)

    def apply_batch(self, chunk_batch):
        """[channels, N] or [steps, channels, N] float32 -> same shape."""
        x = np.asarray(chunk_batch, dtype=np.float32)
        for f in self._filters():
            x = f.apply_batch(x)
        return self.line.apply_host(x)

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """Filters (if enabled) and taps on device-resident batches; the filter stages write to scratch tensors."""
        cur = d_in
        for f in self._filters():
            import torch
            nxt = torch.empty((n_steps, self.channels, self._n), dtype=torch.float32, device=f"cuda:{self.line.device}")
            f.engine.apply_device(cur, nxt, n_steps, stream)
            cur = nxt
        self.line.apply_device(cur, d_out, n_steps, stream)

    def reset(self):
        for f in self._filters():
            f.reset()
        self.line.reset()


class CreateReverb:
    """The reference's experimental, unexported reverb (_EffectReverb.py:5-61): two wet delay lines, HighCut(5000) -> 99
    taps every reverb/100 samples and HighCut(150) -> 49 taps every reverb/50 samples (gains linspace(0.3, 0.01, loops),
    the last one unused), summed.  ``applyreverb(chunk)`` like the reference; works for any chunk size (the reference's
    buffers overflow for chunks longer than about reverb/100 samples)."""

    def __init__(self, time_in_ms=1500, *, channels=1, device=0):
        if config.chunk_size is None or config.sampling_rate is None:
            raise RuntimeError("call config.initialize(sampling_rate, chunk_size) before creating devices")
        self._n, self.channels, self.device = int(config.chunk_size), int(channels), int(device)
        self.reverb_time = time_in_ms
        self.reverb_time_in_samples = int((time_in_ms / 1000) * config.sampling_rate)
        self.lines = []
        for loops, cutoff in ((100, 5000), (50, 150)):
            spacing = self.reverb_time_in_samples // loops
            if spacing < 1:
                raise ValueError("reverb time too short")
            gains = np.linspace(0.3, 0.01, num=loops, dtype="float32")[:loops - 1]
            line = DelayLine([spacing * (k + 1) for k in range(loops - 1)], gains, dry=0.0, chunk_size=self._n,
                             channels=channels, device=device)
            self.lines.append((CreateHighCutFilter(cutoff, channels=channels, device=device), line))
        self.lines[1][1].set_accumulate(True)  # the second line adds onto the first one's output

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        import torch
        scratch = torch.empty((n_steps, self.channels, self._n), dtype=torch.float32, device=f"cuda:{self.device}")
        for hc, line in self.lines:  # stream-ordered: line 0 overwrites d_out, line 1 adds
            hc.engine.apply_device(d_in, scratch, n_steps, stream)
            line.apply_device(scratch, d_out, n_steps, stream)

    def apply_batch(self, chunk_batch):
        import torch
        x = np.ascontiguousarray(chunk_batch, dtype=np.float32)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.channels, self._n):
            raise ValueError(f"expected [steps, {self.channels}, {self._n}], got {x.shape}")
        dev = f"cuda:{self.device}"
        d_in = torch.from_numpy(x).to(dev)
        d_out = torch.empty_like(d_in)
        self.apply_device(d_in, d_out, x.shape[0])
        y = d_out.cpu().numpy()
        return y[0] if squeeze else y

    def applyreverb(self, float32_array_input):
        if self.channels != 1:
            raise ValueError("this device holds several channels; use apply_batch(x[channels, chunk])")
        flat = np.concatenate((float32_array_input,), axis=None)
        if flat.size != self._n:
            raise ValueError(f"chunk has {flat.size} samples, config.chunk_size was {self._n} when this device was created")
        return self.apply_batch(flat.reshape(1, self._n)).reshape(self._n)

    def reset(self):
        for hc, line in self.lines:
            hc.reset()
            line.reset()
