"""Drop-in device classes: CreateHighCutFilter, CreateLowCutFilter, CreateEQ3BandFFT.

Same names, positional arguments, defaults, attributes and ``.apply(chunk) -> float32[N]``
semantics as the reference (pyAudioDspTools/EffectFFTFilter.py:5-151, EffectEQ3BandFFT.py:23-211,
exported at pyAudioDspTools/__init__.py:20-21) - one chunk of latency, zero initial history, a
fresh float32 array per call - but the arithmetic runs in the HIP engine.  Keyword-only extras
(``channels``, ``device``) turn one object into a bank of independent mono channels for
``apply_batch``; they default to the reference's one-object-one-channel behaviour.

Differences that are deliberate (SURVEY.md section 7 "Hard parts"):
  * a wrong-length chunk raises ValueError BEFORE the history is touched (the reference raises the
    same ValueError from numpy after it has already rotated its history, EffectFFTFilter.py:63-71);
  * the caller's array is copied to the GPU, so mutating it after apply() cannot change later
    outputs.  The reference keeps REFERENCES to the caller's arrays as its history
    (EffectFFTFilter.py:63-65 / :139-141, EffectEQ3BandFFT.py:172-174) and re-reads them at the next
    two calls; a caller that overwrites a chunk it has passed - Example4.py:9,18-19 does:
    ``split_data[i] = device.apply(split_data[i])`` on rows of ONE 2-D array - therefore gets, from
    the reference, a stream filtered over (out[k-2], out[k-1], x[k]) instead of (x[k-2], x[k-1], x[k]):
    full-scale different from the FIR stream (tests/golden/kat_inplace.npz records both).  The
    default here is the FIR stream whatever the caller does with its arrays afterwards;
    ``alias_history=True`` (keyword-only, every device) reproduces the reference's behaviour exactly:
    the device keeps references and transforms what they hold at the time of each call;
  * float64 / list inputs are accepted like the reference's ``concatenate(axis=None)`` does, but are
    rounded to float32 on entry.

Non-finite samples: one NaN / Inf sample makes the reference's 3N-point transform - the whole returned
chunk - NaN in the call that takes it and in the two calls after it (kat_nonfinite.npz).  ``apply()``
does the same (a flag per call on the host for numpy chunks, ``adsp_nonfinite_guard`` behind the filter
launch for device-resident chunks).  ``apply_batch`` and the engine-level entry points - which the
reference does not have - let the kernels' arithmetic decide: the overlap-save blocks whose window holds
the sample come out non-finite, always a superset of the FIR's support and a subset of those three
chunks (tests/test_gpu_round6.py pins both statements).
"""
import numpy as np

from . import config
from .design import (TRIM_EPS, FirStream, eq3_composite, eq3_kernels, filter_length, highcut_kernel, lowcut_kernel,
                     reference_spectrum_3n)
from .effects import Effect
from .engine import FirEngine, make_engine


def _is_device_tensor(x):
    """A torch tensor that lives on a GPU (the stand-in for the cupy arrays of the reference's *GPU twins)."""
    return type(x).__module__.split(".")[0] == "torch" and bool(getattr(x, "is_cuda", False))


def _host(a):
    """numpy view / copy of whatever a device keeps as history: numpy arrays as they are, device tensors copied back."""
    return a.detach().cpu().numpy() if _is_device_tensor(a) else a


class _FFTDevice:
    """Shared plumbing of the three devices."""

    def _setup(self, fir_taps, channels, device, alias_history=False):
        n = config.chunk_size
        if n is None or config.sampling_rate is None:
            raise RuntimeError("call config.initialize(sampling_rate, chunk_size) before creating devices")
        self._n = int(n)
        self.channels = int(channels)
        self.filter_length, d = filter_length(self._n)
        self.array_slice_value_start = self._n + (self.filter_length // 2)
        self.array_slice_value_end = self._n - (self.filter_length // 2)
        self.cut_size = np.int16((self.filter_length - 1) / 2)
        self.fir = FirStream(fir_taps, self._n, latency_chunks=1, lookahead=d)
        self.engine = make_engine(self.fir, channels=self.channels, device=device)
        zeros = np.zeros(self._n) if self.channels == 1 else np.zeros((self.channels, self._n))
        self.float32_array_input_1 = zeros
        self.float32_array_input_2 = zeros
        self.float32_array_input_3 = zeros
        self._last_output = None
        self.alias_history = bool(alias_history)
        if self.alias_history and self.channels != 1:
            raise ValueError("alias_history reproduces the reference's one-object-one-channel apply(); it is not available for banks")
        self._zeros32 = np.zeros(self._n, np.float32)
        self._bad_calls = [False, False, False]   # non-finite sample seen in call k, k-1, k-2 (host chunks)
        self._guard_flags = None                  # the same ring on the GPU (device-resident chunks): uint32 [1][3]
        self._calls = 0

    def _rotate(self, x):
        self.float32_array_input_3 = self.float32_array_input_2
        self.float32_array_input_2 = self.float32_array_input_1
        self.float32_array_input_1 = x

    def _nonfinite(self, x32):
        """True when the float32 vector holds a NaN / Inf: x . 0 is NaN exactly then (finite * 0 = 0, Inf * 0 = NaN * 0 = NaN) - one
        BLAS dot (about a microsecond at N = 4096) instead of an elementwise pass and a reduction."""
        return bool(np.isnan(np.dot(x32, self._zeros32)))

    def apply(self, float32_array_input):
        """One chunk in, the previous chunk (filtered) out: float32 array of config.chunk_size samples."""
        if self.channels != 1:
            raise ValueError("this device holds several channels; use apply_batch(x[channels, chunk])")
        if _is_device_tensor(float32_array_input):
            return self._apply_device_tensor(float32_array_input)
        flat = np.concatenate((float32_array_input,), axis=None)  # same flattening as the reference
        if flat.size != self._n:
            raise ValueError(f"operands could not be broadcast together: chunk has {flat.size} samples, "
                             f"config.chunk_size was {self._n} when this device was created")
        x = np.ascontiguousarray(flat, dtype=np.float32)
        if self.alias_history:
            y = self._apply_aliased_host(x)
        else:
            y = self.engine.apply_host(x.reshape(1, self._n)).reshape(self._n)
            earlier = self._bad_calls[:2]
            if self._guard_flags is not None:  # earlier chunks came as device tensors: their findings live on the GPU
                on_gpu = self._guard_flags_host()
                earlier = [earlier[0] or bool(on_gpu[0]), earlier[1] or bool(on_gpu[1])]
                self._guard_flags = None
            self._bad_calls = [self._nonfinite(x), earlier[0], earlier[1]]
            if any(self._bad_calls):
                y = np.full(self._n, np.nan, np.float32)  # the reference's 3N transform of a window holding the sample: all NaN
        self._calls += 1
        self._rotate(float32_array_input)
        self._last_output = y
        return y

    def _history_as_float32(self, h):
        flat = np.concatenate((_host(h),), axis=None)
        if flat.size != self._n:
            raise ValueError(f"operands could not be broadcast together: a history chunk now has {flat.size} samples")
        return np.ascontiguousarray(flat, dtype=np.float32)

    def _apply_aliased_host(self, x):
        """alias_history: the reference's window is whatever the arrays it REFERENCES hold now (EffectFFTFilter.py:63-68).  The engine
        starts from zero history and takes the three chunks as three steps of one launch sequence; the third step's output depends on
        exactly those 3N samples (the kept slice never sees the wrap-around), so it is the reference's result for them."""
        window = np.stack([self._history_as_float32(self.float32_array_input_2), self._history_as_float32(self.float32_array_input_1), x])
        self.engine.reset()
        y = self.engine.apply_host(window.reshape(3, 1, self._n))[2].reshape(self._n)
        if self._nonfinite(window.reshape(-1)[:self._n]) or self._nonfinite(window[1]) or self._nonfinite(x):
            y = np.full(self._n, np.nan, np.float32)
        return np.ascontiguousarray(y)

    def _guard_flags_host(self):
        """[call k, k-1, k-2] findings of the device-side flag ring, in the order of _bad_calls."""
        f = self._guard_flags.cpu().numpy().reshape(3)
        newest = (self._calls - 1) % 3
        return [int(f[(newest - j) % 3]) for j in range(3)]

    def _apply_device_tensor(self, x):
        """The *GPU twins' call (EffectFFTFilterGPU.py:54-78, EffectEQ3BandFFTGPU.py:161-215: device array in, device array out - Example4.py:19)
        with a torch tensor where the reference takes a cupy array: no host copy in either direction, the launch goes on torch's current
        stream of the tensor's device and the result is a fresh float32 tensor there."""
        import torch
        from . import _capi
        from .engine import _ptr
        if x.numel() != self._n:
            raise ValueError(f"operands could not be broadcast together: chunk has {x.numel()} samples, "
                             f"config.chunk_size was {self._n} when this device was created")
        if x.device.index != self.engine.device:
            raise ValueError(f"the chunk lives on GPU {x.device.index}, this device was created on GPU {self.engine.device}")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        xin = x.detach().reshape(self._n).to(torch.float32).contiguous()
        if self.alias_history:
            # the reference's window: what the referenced arrays hold NOW (Example4's loop has overwritten them with outputs)
            def on_device(h):
                if _is_device_tensor(h):
                    return h.detach().reshape(self._n).to(device=x.device, dtype=torch.float32)
                return torch.from_numpy(self._history_as_float32(h)).to(x.device)
            window = torch.stack([on_device(self.float32_array_input_2), on_device(self.float32_array_input_1), xin]).contiguous()
            out3 = torch.empty_like(window)
            self.engine.reset()
            self.engine.apply_device(window.reshape(3, 1, self._n), out3.reshape(3, 1, self._n), 3, stream)
            y = out3[2]
            # the three chunks ARE the window: scan each into its slot (flags rewritten at every call), the last one poisons y
            flags = torch.zeros((1, 3), dtype=torch.int32, device=x.device)
            scratch = torch.empty(self._n, device=x.device, dtype=torch.float32)
            lib = _capi.load()
            for slot in range(3):
                _capi.check(lib.adsp_nonfinite_guard(self.engine.device, _ptr(window[slot]), _ptr(y if slot == 2 else scratch), 1, self._n,
                                                     _ptr(flags), slot, _ptr(stream)))
        else:
            y = torch.empty(self._n, device=x.device, dtype=torch.float32)
            self.engine.apply_device(xin, y, 1, stream)
            if self._guard_flags is None:
                self._guard_flags = torch.zeros((1, 3), dtype=torch.int32, device=x.device)
                if any(self._bad_calls):  # earlier chunks came as numpy arrays: carry their findings over
                    host = [0, 0, 0]
                    for j in (1, 2):      # call k-1 -> _bad_calls[0], call k-2 -> _bad_calls[1]
                        host[(self._calls - j) % 3] = int(self._bad_calls[j - 1])
                    self._guard_flags.copy_(torch.tensor([host], dtype=torch.int32))
                self._bad_calls = [False, False, False]
            _capi.check(_capi.load().adsp_nonfinite_guard(self.engine.device, _ptr(xin), _ptr(y), 1, self._n, _ptr(self._guard_flags),
                                                          self._calls % 3, _ptr(stream)))
        self._calls += 1
        self._rotate(x)
        self._last_output = y
        return y

    def apply_batch(self, chunk_batch):
        """[channels, N] (one step) or [steps, channels, N] float32 -> same shape."""
        x = np.asarray(chunk_batch)
        y = self.engine.apply_host(x)
        self._rotate(x if x.ndim == 2 else x[-1])
        self._last_output = y if y.ndim == 2 else y[-1]
        return y

    def reset(self):
        self.engine.reset()
        zeros = np.zeros(self._n) if self.channels == 1 else np.zeros((self.channels, self._n))
        self.float32_array_input_1 = self.float32_array_input_2 = self.float32_array_input_3 = zeros
        self._last_output = None
        self._bad_calls, self._guard_flags, self._calls = [False, False, False], None, 0

    def _concatenated_inputs(self):
        """The 3N-sample buffer the reference transforms: chunks k-2, k-1, k flattened (EffectFFTFilter.py:67-68)."""
        return np.concatenate((_host(self.float32_array_input_3), _host(self.float32_array_input_2), _host(self.float32_array_input_1)), axis=None)

    def _cut_filter_filtered_signal(self):
        # EffectFFTFilter.py:39 (zeros(3N) until the first apply), :67-73 (afterwards the sliced inverse transform: N complex values
        # whose real part, cast to float32, is what apply returned).  Read-only here: the arithmetic happened on the GPU, the imaginary
        # part - rounding noise of the reference's complex transform - is exactly zero.
        if self._last_output is None:
            return np.zeros(self._n * 3)
        return np.asarray(_host(self._last_output)).astype(np.complex128)


class CreateHighCutFilter(_FFTDevice):
    """FFT high-cut (low-pass) device.  cutoff_frequency defaults to 8000 like the reference
    (EffectFFTFilter.py:18)."""

    def __init__(self, cutoff_frequency=8000, *, channels=1, device=0, alias_history=False):
        self.fS = config.sampling_rate
        self.fH = cutoff_frequency
        taps = highcut_kernel(self.fH, self.fS, config.chunk_size)
        self._setup(taps, channels, device, alias_history)

    @property
    def sinc_filter(self):
        """3N-point complex128 spectrum, as the reference exposes it (EffectFFTFilter.py:45-47)."""
        return reference_spectrum_3n(self.fir.taps, self._n)

    @property
    def filtered_signal(self):
        """The reference's inspectable work buffer (EffectFFTFilter.py:39, :67-73), read-only."""
        return self._cut_filter_filtered_signal()


class CreateLowCutFilter(_FFTDevice):
    """FFT low-cut (high-pass) device.  cutoff_frequency defaults to 160 (EffectFFTFilter.py:91)."""

    def __init__(self, cutoff_frequency=160, *, channels=1, device=0, alias_history=False):
        self.fS = config.sampling_rate
        self.fH = cutoff_frequency
        taps = lowcut_kernel(self.fH, self.fS, config.chunk_size)
        self._setup(taps, channels, device, alias_history)

    @property
    def sinc_filter(self):
        return reference_spectrum_3n(self.fir.taps, self._n)

    @property
    def filtered_signal(self):
        """The reference's inspectable work buffer (EffectFFTFilter.py:115, :143-149), read-only."""
        return self._cut_filter_filtered_signal()


class CreateEQ3BandFFT(_FFTDevice):
    """3-band FFT EQ; six positional arguments, no defaults (EffectEQ3BandFFT.py:47)."""

    def __init__(self, lowshelf_frequency, lowshelf_db, midband_frequency, midband_db, highshelf_frequency,
                 highshelf_db, *, channels=1, device=0, alias_history=False):
        self.fS = config.sampling_rate
        self.fH_highshelf = highshelf_frequency
        self.highshelf_db = highshelf_db
        self.fH_lowshelf = lowshelf_frequency
        self.lowshelf_db = lowshelf_db
        self.fH_midband = midband_frequency
        self.midband_db = midband_db
        self._bands = eq3_kernels(lowshelf_frequency, midband_frequency, highshelf_frequency, self.fS,
                                  config.chunk_size)
        taps = eq3_composite(lowshelf_frequency, lowshelf_db, midband_frequency, midband_db, highshelf_frequency,
                             highshelf_db, self.fS, config.chunk_size)
        self._setup(taps, channels, device, alias_history)

    @property
    def filtered_signal(self):
        """EffectEQ3BandFFT.py:147: zeros(3N) - the reference's apply works on locals and never touches it again."""
        return np.zeros(self._n * 3)

    @property
    def original_signal(self):
        """EffectEQ3BandFFT.py:148, :175-176: the three most recent chunks flattened (zeros(3N) before the first apply).  Read-only."""
        return self._concatenated_inputs()

    @property
    def sinc_filter_highshelf(self):
        return reference_spectrum_3n(self._bands["highshelf"], self._n)

    @property
    def sinc_filter_lowshelf(self):
        return reference_spectrum_3n(self._bands["lowshelf"], self._n)

    @property
    def sinc_filter_mid_lowpass(self):
        return reference_spectrum_3n(self._bands["mid_lowpass"], self._n)

    @property
    def sinc_filter_mid_highpass(self):
        return reference_spectrum_3n(self._bands["mid_highpass"], self._n)


# The reference's cupy twins (EffectFFTFilterGPU.py, EffectEQ3BandFFTGPU.py) have the same surface; a GPU-resident chunk (a torch tensor,
# where the reference takes a cupy array) stays on the GPU through apply() - with either name.
CreateHighCutFilterGPU = CreateHighCutFilter
CreateLowCutFilterGPU = CreateLowCutFilter
CreateEQ3BandFFTGPU = CreateEQ3BandFFT


def fuse(*devices, channels=None, device=0, ring_slots=0, trim=None, optimize_for="stream"):
    """Series connection of FFT devices as ONE engine (config 5: LowCut -> EQ3 -> HighCut).

    The result computes, in a single kernel per step, what feeding each device's output into the
    next one's apply() computes in the reference: one FIR of summed length, latency = number of
    devices chunks.  A stateless effect (effects.CreateSoftClipper, ...) may follow the last device; it is applied to
    the kernel's output registers (no extra pass).

    `trim`: the composite kernel's negligible end taps are left out (FirStream.trimmed; default design.TRIM_EPS = 1e-8 of
    sum|taps|, i.e. a worst-case output change of ~6e-8 of full scale for the config-5 chain); 0 keeps every tap."""
    devices = list(devices)
    effect = devices.pop() if isinstance(devices[-1], Effect) else None
    if not devices or any(isinstance(dev, Effect) for dev in devices):
        raise ValueError("fuse() takes FFT devices, optionally followed by ONE stateless effect at the end")
    fir = devices[0].fir
    for dev in devices[1:]:
        fir = fir.then(dev.fir)
    if len(devices) > 1 and trim != 0:
        fir = fir.trimmed(TRIM_EPS if trim is None else trim)
    ch = devices[0].channels if channels is None else channels
    engine = FirEngine(fir, channels=ch, device=device, ring_slots=ring_slots, optimize_for=optimize_for)
    if effect is not None:
        engine.set_epilogue(effect)
    return engine
