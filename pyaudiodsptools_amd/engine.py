"""FirEngine: Python handle on one libadsp engine (C channels of one streaming FIR on one GPU).

This is the batched form of the reference's ``.apply()``: where the reference runs one Python
object per mono channel and one numpy FFT pipeline per chunk (EffectFFTFilter.py:49-75), an engine
holds the history of C channels in HBM and filters a whole ``[steps, C, N]`` batch per launch.
"""
import ctypes

import numpy as np

from . import _capi
from .design import (PCM16_GAIN, FirStream, choose_uniform_block, engine_spectrum, fits_one_transform, overlap_save_geometry, partition,
                     partition_uniform)

_FORMATS = {"f32": (_capi.ADSP_FORMAT_F32, np.float32), "s16": (_capi.ADSP_FORMAT_S16, np.int16),
            "s16_f64": (_capi.ADSP_FORMAT_S16_F64, np.int16)}  # int16 samples, float64 arithmetic (the exact-FFT engines)


def _ptr(x):
    """Device/host address of a torch tensor or numpy array, the handle of a torch stream, or a raw integer address / handle."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if hasattr(x, "data_ptr"):  # torch tensor (plumbing only)
        return ctypes.c_void_p(x.data_ptr())
    if hasattr(x, "cuda_stream"):  # torch.cuda.Stream: the raw hipStream_t
        return ctypes.c_void_p(x.cuda_stream)
    if isinstance(x, np.ndarray):
        return ctypes.c_void_p(x.ctypes.data)
    raise TypeError(f"cannot take the address of {type(x)}")


class FirEngine:
    def __init__(self, fir: FirStream, channels=1, device=0, ring_slots=0, fft_mult=0, sample_format="f32",
                 optimize_for="stream"):
        self._lib = _capi.load()
        self._h = ctypes.c_void_p(None)
        self.fir = fir
        self.fft_mult = fft_mult if fft_mult != int(fft_mult) else int(fft_mult)
        if sample_format not in _FORMATS:
            raise ValueError("sample_format must be 'f32', 's16' or 's16_f64'")
        self.sample_format = sample_format
        self._fmt_code, self.dtype = _FORMATS[sample_format]
        # int16 engines: (float)x in, (int16)trunc(y) out; the reference's /32768 and *32767 live in the spectrum.
        # float64 int16 engines apply both conversions to the letter inside the kernel: no gain in the spectrum
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
        self.epilogue = None

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
    def set_fir(self, fir: FirStream, stream=None, live=False):
        """Change the filter without touching the history (same geometry required).  live=True: stream-ordered
        (adsp_set_spectrum_async) - steps already queued on `stream` finish with the old filter, later ones use the new
        one, nothing is synchronised; otherwise the device is drained first (set-up path)."""
        geo = overlap_save_geometry(fir, self.fft_mult, self.optimize_for)
        if getattr(self, "geometry", geo) != geo:
            raise ValueError("new filter needs a different transform geometry; create a new engine")
        self.fir = fir
        self.spectrum = engine_spectrum(fir, geo, self.gain)
        if self.sample_format == "s16_f64":
            if live:
                raise ValueError("float64 engines change their filter through the set-up path only")
            spec64 = engine_spectrum(fir, geo, 1.0, np.float64)
            _capi.check(self._lib.adsp_set_spectrum_f64(self._h, _ptr(spec64), spec64.size // 2))
        elif live:
            _capi.check(self._lib.adsp_set_spectrum_async(self._h, _ptr(self.spectrum), self.spectrum.size // 2, _ptr(stream)))
        else:
            self.upload_spectrum(self.spectrum)
        # taps at negative circular indices: lets the kernel skip the part of the window that feeds discarded outputs
        _capi.check(self._lib.adsp_set_kernel_reach(self._h, max(0, -geo.shift)))

    def upload_spectrum(self, spectrum_f32, reach=None):
        """Set a spectrum computed elsewhere (e.g. received by broadcast).  Every spectrum upload forgets the kernel-reach
        hint; `reach` = taps at negative circular indices of THIS spectrum's kernel restores it (None: fetch whole windows)."""
        spec = np.ascontiguousarray(spectrum_f32, dtype=np.float32)
        _capi.check(self._lib.adsp_set_spectrum(self._h, _ptr(spec), spec.size // 2))
        if reach is not None:
            _capi.check(self._lib.adsp_set_kernel_reach(self._h, int(reach)))

    @property
    def real_spectrum(self):
        """True when the engine runs the real-spectrum stage (symmetric kernel centred on circular index 0)."""
        flag = ctypes.c_int(0)
        _capi.check(self._lib.adsp_spectrum_is_real(self._h, ctypes.byref(flag)))
        return bool(flag.value)

    def upload_spectrum_device(self, d_spectrum, n_bins, stream=None, reach=None):
        _capi.check(self._lib.adsp_set_spectrum_device(self._h, _ptr(d_spectrum), int(n_bins), _ptr(stream)))
        if reach is not None:
            _capi.check(self._lib.adsp_set_kernel_reach(self._h, int(reach)))

    def get_spectrum(self):
        """The interleaved float32 spectrum the engine's tables were last built from (after a broadcast: what the collective
        left in this engine's device memory)."""
        out = np.empty(2 * (self.geometry.fft_size // 2 + 1), np.float32)
        _capi.check(self._lib.adsp_get_spectrum(self._h, _ptr(out), out.size // 2))
        return out

    def bcast_rank(self, unique_id, rank, world, root=0):
        """adsp_bcast_spectrum_rank: this process is rank `rank` of `world` (one process per GPU); every rank calls it with
        the 128 bytes rank 0 drew with rccl_unique_id().  Afterwards every engine runs the root's filter."""
        uid = bytes(unique_id)
        if len(uid) != _capi.ADSP_RCCL_UNIQUE_ID_BYTES:
            raise ValueError(f"unique_id must be {_capi.ADSP_RCCL_UNIQUE_ID_BYTES} bytes")
        _capi.check(self._lib.adsp_bcast_spectrum_rank(self._h, uid, int(rank), int(world), int(root)))
        self.spectrum = self.get_spectrum()

    def set_accumulate(self, mode=True):
        """0/False overwrite the output buffer; 1/True add to what it holds (later parts of a partitioned FIR, a mix
        bus); 2 add and clip the sum to [-1, 1] (the last engine of a MixSignals bus, Utility.py:51-72)."""
        _capi.check(self._lib.adsp_set_accumulate(self._h, int(mode)))

    def set_epilogue(self, effect=None):
        """Fuse a stateless effect (effects.Effect) onto the kernel's output: later applies return effect(filter(x)).
        None removes it.  float32 engines only."""
        op, (p0, p1, p2) = (effect.op, effect.params()) if effect is not None else (_capi.EFFECT_NONE, (0.0, 0.0, 0.0))
        _capi.check(self._lib.adsp_set_epilogue(self._h, int(op), float(p0), float(p1), float(p2)))
        self.epilogue = effect

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

    def get_epilogue_state(self):
        """What a fused effect adds to a checkpoint (the tremolo's LFO position; 0 for stateless effects)."""
        v = ctypes.c_longlong(0)
        _capi.check(self._lib.adsp_get_epilogue_state(self._h, ctypes.byref(v)))
        return v.value

    def set_epilogue_state(self, state):
        _capi.check(self._lib.adsp_set_epilogue_state(self._h, int(state)))

    def set_state(self, history):
        h = np.ascontiguousarray(history, dtype=self.dtype)
        if h.shape != (self.geometry.history_chunks, self.channels, self.chunk_size):
            raise ValueError(f"state must have shape {(self.geometry.history_chunks, self.channels, self.chunk_size)}")
        _capi.check(self._lib.adsp_set_state(self._h, _ptr(h)))

    # -- apply --------------------------------------------------------------------------------
    def apply_host(self, x, out=None):
        """x: host array [steps, C, N] (or [C, N]) of the engine's sample type -> same shape; a fresh array like the reference's apply
        (EffectFFTFilter.py:75), or `out` (C-contiguous, same shape and type: a gigabyte-sized batch need not be allocated and
        page-faulted per call).  Batches beyond a few megabytes move in slabs through pinned staging, copies overlapped with the
        kernels (adsp_apply_host)."""
        if self.sample_format != "f32" and np.asarray(x).dtype != np.int16:
            raise TypeError("this engine filters int16 PCM; pass an int16 array")
        x = np.ascontiguousarray(x, dtype=self.dtype)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.channels, self.chunk_size):
            raise ValueError(f"expected [steps, {self.channels}, {self.chunk_size}], got {x.shape}")
        if out is None:
            res = np.empty_like(x)
        else:
            res = out[None] if squeeze and out.ndim == 2 else out
            if res.shape != x.shape or res.dtype != x.dtype or not res.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous array of the input's shape and type")
        _capi.check(self._lib.adsp_apply_host(self._h, _ptr(x), _ptr(res), x.shape[0]))
        return res[0] if squeeze else res

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """Asynchronous, device-resident [n_steps, C, N] float32 buffers (torch tensors or addresses)."""
        _capi.check(self._lib.adsp_apply_device(self._h, _ptr(d_in), _ptr(d_out), int(n_steps), _ptr(stream)))

    def ring_acquire(self, stream=None):
        """Device address of the ring slot the producer must fill with the next [C, N] batch.  Pass the stream the
        producer runs on when steps are issued on several streams: it is ordered after the kernels that still read the
        slot (adsp_ring_acquire_stream); without it the host blocks until they have finished."""
        p = ctypes.c_void_p(None)
        if stream is None:
            _capi.check(self._lib.adsp_ring_acquire(self._h, ctypes.byref(p)))
        else:
            _capi.check(self._lib.adsp_ring_acquire_stream(self._h, ctypes.byref(p), _ptr(stream)))
        return p.value

    def ring_set_pipeline(self, depth=2):
        """depth 2: the library runs ring step k on its own stream k % 2 (consecutive launches overlap); depth 3: the steps RIDE A LIVE
        SESSION the library starts, feeds and stops by itself (one persistent launch, history on chip: adsp_live_*) - available where a
        session can hold the engine (AdspError otherwise); "auto": 3 where possible, else 2.  The caller keeps one stream for its
        producers (ring_acquire(stream) -> fill the slot on that stream -> apply_ring(out, stream)) and calls ring_join(stream) before
        anything reads outputs.  depth 1: the default (every step a launch on the caller's stream).  Returns the depth in force."""
        if depth == "auto":
            try:
                _capi.check(self._lib.adsp_ring_set_pipeline(self._h, 3))
                return 3
            except _capi.AdspError:
                depth = 2
        _capi.check(self._lib.adsp_ring_set_pipeline(self._h, int(depth)))
        return int(depth)

    def ring_join(self, stream=None):
        _capi.check(self._lib.adsp_ring_join(self._h, _ptr(stream)))

    def ring_reset_order(self):
        """Drain the device and forget the cross-stream ordering events of the ring (before and after hipGraph capture)."""
        _capi.check(self._lib.adsp_ring_reset_order(self._h))

    # resident ring launches (include/adsp.h): the producer publishes steps, one consumer launch covers many of them
    def ring_produce_begin(self, stream=None):
        p = ctypes.c_void_p(None)
        _capi.check(self._lib.adsp_ring_produce_begin(self._h, ctypes.byref(p), _ptr(stream)))
        return p.value

    def ring_produce_end(self, stream=None):
        _capi.check(self._lib.adsp_ring_produce_end(self._h, _ptr(stream)))

    def apply_ring_resident(self, d_out, n_steps, stream=None):
        """d_out [n_steps, C, N]: consumes the next n_steps ring steps in ONE launch; step k's workgroups start when the
        producer has published it (ring_produce_begin / _end), which may happen after this call."""
        _capi.check(self._lib.adsp_apply_ring_resident(self._h, _ptr(d_out), int(n_steps), _ptr(stream)))

    def ring_resident_timeout(self, milliseconds):
        _capi.check(self._lib.adsp_ring_resident_timeout(self._h, float(milliseconds)))

    def ring_resident_timed_out(self):
        """True (once) when a workgroup of a resident launch gave up waiting for its step; synchronises on the flag."""
        v = ctypes.c_int(0)
        _capi.check(self._lib.adsp_ring_resident_status(self._h, ctypes.byref(v)))
        return bool(v.value)

    # live sessions (include/adsp.h): ONE persistent launch consumes ring steps as they are published
    def live_configure(self, step_timeout_ms=1000.0, load_mode=2):
        _capi.check(self._lib.adsp_live_configure(self._h, float(step_timeout_ms), int(load_mode)))

    def live_start(self, d_out, out_slots, max_steps, stream):
        """d_out [out_slots, C, N]: step s of the session writes slot s % out_slots.  Use an explicitly created stream."""
        _capi.check(self._lib.adsp_live_start(self._h, _ptr(d_out), int(out_slots), int(max_steps), _ptr(stream)))

    def live_slot(self):
        """Device address of the ring slot the next step's [C, N] batch goes to (AdspError 'ring full' while the session lags)."""
        p = ctypes.c_void_p(None)
        _capi.check(self._lib.adsp_live_slot(self._h, ctypes.byref(p)))
        return p.value

    def live_publish(self, stream=None):
        """Publish every slot handed out since the last publication: with a stream, a one-lane kernel behind the commands that
        filled them; without, a plain host store (the data must already be complete and visible)."""
        if stream is None:
            _capi.check(self._lib.adsp_live_publish_host(self._h))
        else:
            _capi.check(self._lib.adsp_live_publish_stream(self._h, _ptr(stream)))

    def live_publish_run(self, n_steps, stream=None):
        """A data-less producer in a native loop: publish the next n_steps slots one by one (bench / soak)."""
        _capi.check(self._lib.adsp_live_publish_run(self._h, int(n_steps), 0 if stream is None else 1, _ptr(stream)))

    def live_progress(self):
        v = ctypes.c_uint(0)
        _capi.check(self._lib.adsp_live_progress(self._h, ctypes.byref(v)))
        return v.value

    def live_wait(self, steps, timeout_ms=10000.0):
        _capi.check(self._lib.adsp_live_wait(self._h, int(steps), float(timeout_ms)))

    def live_stop(self):
        """End the session (after every published step is consumed); returns the steps consumed."""
        v = ctypes.c_uint(0)
        _capi.check(self._lib.adsp_live_stop(self._h, ctypes.byref(v)))
        return v.value

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


def broadcast_filter(engines, root=0):
    """adsp_bcast_spectrum: every engine of the list (ONE per GPU, all in this process, same geometry) takes over the
    filter of engines[root] through an RCCL broadcast inside libadsp - no torch, no rendezvous."""
    lib = _capi.load()
    arr = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    if all(isinstance(e, UpolsFirEngine) for e in engines):  # kernels longer than one transform: adsp_upols_bcast_spectra
        _capi.check(lib.adsp_upols_bcast_spectra(arr, len(engines), int(root)))
        for e in engines:
            e.fir = engines[root].fir
        return
    if any(isinstance(e, UpolsFirEngine) for e in engines):
        raise ValueError("engines of different kinds cannot share a filter")
    _capi.check(lib.adsp_bcast_spectrum(arr, len(engines), int(root)))
    for e in engines:
        e.fir, e.spectrum = engines[root].fir, e.get_spectrum()  # each engine's own copy, as the collective left it


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 of a one-process-per-GPU job hands them to the other ranks)."""
    buf = ctypes.create_string_buffer(_capi.ADSP_RCCL_UNIQUE_ID_BYTES)
    _capi.check(_capi.load().adsp_rccl_unique_id(buf))
    return buf.raw


class ClockProbe:
    """Shader clock while a workload runs: a one-lane kernel on `stream` (a side stream) counts shader cycles over
    `microseconds` of the constant 100 MHz clock; read() waits for it and returns MHz."""

    def __init__(self, device, microseconds, stream):
        self._lib, self._device, self._stream = _capi.load(), int(device), stream
        self._res = ctypes.c_void_p(None)
        _capi.check(self._lib.adsp_clock_probe_launch(self._device, float(microseconds), _ptr(stream), ctypes.byref(self._res)))

    def read(self):
        mhz = ctypes.c_double(0.0)
        _capi.check(self._lib.adsp_clock_probe_read(self._device, _ptr(self._stream), self._res, ctypes.byref(mhz)))
        self._res = ctypes.c_void_p(None)
        return mhz.value

    def __del__(self):
        # A probe nobody read keeps its 16 bytes of pinned result memory: freeing it means waiting for the probe kernel (and for whatever
        # is queued in front of it on its stream), and a destructor - run by the garbage collector at an arbitrary point - must not block
        # on the GPU.  bench.py reads every probe it launches.
        self._res = None


def rccl_finalize():
    """adsp_rccl_finalize: destroy every RCCL communicator libadsp has built (no broadcast may be in flight) - and forget the unique
    ids this process took part with (dist.exchange_unique_id caches them per id file): a communicator is gone with its id, the next
    broadcast of a job that re-forms draws a fresh one."""
    _capi.check(_capi.load().adsp_rccl_finalize())
    from . import dist
    dist.forget_unique_ids()


def rccl_version():
    v = ctypes.c_int(0)
    _capi.check(_capi.load().adsp_rccl_version(ctypes.byref(v)))
    return v.value


class ExactFirEngine:
    """Exact mode (adsp_exact_*): the same streaming FIR as a float64 direct sum on the GPU.  For int16 PCM it applies
    the reference's conversions to the letter and returns the int16 stream the reference writes (Utility.py:236-237,
    :306) - what the float32 FFT engines can only do to within one LSB, because the export truncates; for float32 it
    is the on-device ground truth of the parity tests.  O(taps) per sample: files and checks, not the hot path."""

    def __init__(self, fir: FirStream, channels=1, device=0, sample_format="f32"):
        self._lib = _capi.load()
        self._h = ctypes.c_void_p(None)
        if sample_format not in _FORMATS:
            raise ValueError("sample_format must be 'f32' or 's16'")
        self.fir, self.sample_format = fir, sample_format
        code, self.dtype = _FORMATS[sample_format]
        self.chunk_size, self.channels, self.device = int(fir.chunk_size), int(channels), int(device)
        taps = np.ascontiguousarray(fir.taps, dtype=np.float64)
        cfg = _capi.AdspExactConfig(self.device, self.chunk_size, self.channels, len(taps), int(fir.delay), code)
        _capi.check(self._lib.adsp_exact_create(ctypes.byref(cfg), _ptr(taps), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.adsp_exact_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _capi.check(self._lib.adsp_exact_reset(self._h))

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """Device-resident [n_steps, C, N] buffers of the engine's sample type, not aliased; asynchronous."""
        _capi.check(self._lib.adsp_exact_apply_device(self._h, _ptr(d_in), _ptr(d_out), int(n_steps), _ptr(stream)))

    def apply_host(self, x):
        if self.sample_format == "s16" and np.asarray(x).dtype != np.int16:
            raise TypeError("this engine filters int16 PCM; pass an int16 array")
        x = np.ascontiguousarray(x, dtype=self.dtype)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.channels, self.chunk_size):
            raise ValueError(f"expected [steps, {self.channels}, {self.chunk_size}], got {x.shape}")
        out = np.empty_like(x)
        _capi.check(self._lib.adsp_exact_apply_host(self._h, _ptr(x), _ptr(out), x.shape[0]))
        return out[0] if squeeze else out


class PartitionedFirEngine:
    """A FIR too long for one 32768-point transform (e.g. the reference's Example4: chunk 88200, 44099 taps), run as
    P ordinary engines over the same input - one per slice of the kernel, each with the slice's extra delay - whose
    partial outputs are summed.  Same call surface as FirEngine for the float32 paths."""

    def __init__(self, fir: FirStream, channels=1, device=0, max_taps=14336, optimize_for="stream"):
        self.fir = fir
        self.parts = partition(fir, max_taps)
        self.engines = [FirEngine(p, channels=channels, device=device, optimize_for=optimize_for) for p in self.parts]
        for eng in self.engines[1:]:
            eng.set_accumulate(True)
        self.channels, self.chunk_size, self.device = int(channels), int(fir.chunk_size), int(device)
        self.geometry = self.engines[0].geometry
        self.sample_format = "f32"
        self.epilogue = None
        self._lib = _capi.load()

    def set_epilogue(self, effect=None):
        """The partial sums must be complete before a non-linear effect: it runs as one extra in-place elementwise pass.  A tremolo
        starts its LFO at table index 0 here and advances with every chunk the engine filters, every channel in step."""
        self.epilogue = effect
        if effect is not None and effect.op == _capi.EFFECT_TREMOLO:
            import copy
            self._lfo = copy.copy(effect)  # the engine's own LFO position (the effect object stays the caller's)
            self._lfo.reset()

    def close(self):
        for e in self.engines:
            e.close()

    def reset(self):
        for e in self.engines:
            e.reset()
        if self.epilogue is not None and self.epilogue.op == _capi.EFFECT_TREMOLO:
            self._lfo.reset()

    def _run_epilogue(self, d_out, n_steps, stream):
        p0, p1, p2 = (float(v) for v in self.epilogue.params())
        if self.epilogue.op == _capi.EFFECT_TREMOLO:  # per step: all channels' chunks start at the same table index
            base = d_out.data_ptr() if hasattr(d_out, "data_ptr") else int(d_out)
            plane = self.channels * self.chunk_size * 4
            for k in range(int(n_steps)):
                phase = self._lfo._phase(self.chunk_size)
                _capi.check(self._lib.adsp_tremolo_rows_device(self.device, p0, p1, int(p2), int(phase), _ptr(base + k * plane), _ptr(base + k * plane),
                                                               self.channels, self.chunk_size, _ptr(stream)))
            return
        n = int(n_steps) * self.channels * self.chunk_size
        _capi.check(self._lib.adsp_effect_device(self.device, self.epilogue.op, p0, p1, p2, 0, _ptr(d_out), _ptr(d_out), n, _ptr(stream)))

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        for e in self.engines:  # stream-ordered: part 0 overwrites, the others add
            e.apply_device(d_in, d_out, n_steps, stream)
        if self.epilogue is not None:
            self._run_epilogue(d_out, n_steps, stream)

    def apply_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        squeeze = x.ndim == 2
        # host path: each part through its own staging buffers, summed in float64 and rounded once
        acc = None
        for e in self.engines:
            e.set_accumulate(False)
            y = e.apply_host(x).astype(np.float64)
            acc = y if acc is None else acc + y
        for e in self.engines[1:]:
            e.set_accumulate(True)
        out = acc.astype(np.float32)
        if self.epilogue is None:
            return out
        if self.epilogue.op != _capi.EFFECT_TREMOLO:
            return self.epilogue.apply(out, device=self.device)
        import torch  # plumbing: a device copy of the batch for the row-wise pass
        d = torch.from_numpy(out[None] if squeeze else out).cuda(self.device)
        self._run_epilogue(d, d.shape[0], torch.cuda.current_stream(self.device).cuda_stream)
        res = d.cpu().numpy()
        return res[0] if squeeze else res

    def synchronize(self, stream=None):
        self.engines[0].synchronize(stream)


class UpolsFirEngine:
    """A FIR longer than one transform as ONE uniformly partitioned engine (adsp_upols_*, csrc/adsp_upols.hip): every input block of
    B samples (`block`: 8192 or 16384, default design.choose_uniform_block: the larger one where the delay allows it and a call has
    blocks enough to fill the chip several times) is transformed once, its spectrum kept in a frequency-domain delay line in HBM, and an output block is one
    inverse transform of sum_p X_{b-p} H_p - two launches per call instead of PartitionedFirEngine's full engine pass per kernel
    slice.  float32 or int16 batches, a stateless effect fused on the output registers.  The reference shape: Example4.py:5 /
    ModuleTestsGPU.py:35 (chunk 88200: 44 099 / 88 197 taps)."""

    @staticmethod
    def block_sizes():
        lib = _capi.load()
        sizes = (ctypes.c_int * 8)()
        return [int(v) for v in sizes[:lib.adsp_upols_block_sizes(sizes, 8)]]

    def __init__(self, fir: FirStream, channels=1, device=0, sample_format="f32", max_steps=1, optimize_for="stream", block=None,
                 partition=None):
        """`partition`: design.partition_uniform(fir, block, gain) computed by the caller (make_engine does, to test the preconditions)."""
        self._lib = _capi.load()
        self._h = ctypes.c_void_p(None)
        if sample_format not in ("f32", "s16"):
            raise ValueError("sample_format must be 'f32' or 's16'")
        self.fir, self.sample_format = fir, sample_format
        self._fmt_code, self.dtype = _FORMATS[sample_format]
        self.gain = PCM16_GAIN if sample_format == "s16" else 1.0
        self.channels, self.chunk_size, self.device = int(channels), int(fir.chunk_size), int(device)
        sizes = self.block_sizes()
        if block is None:
            block = choose_uniform_block(fir, self.channels, sizes)
        if int(block) not in sizes:
            raise ValueError(f"block {block}: this build partitions into blocks of {sizes} samples")
        self.block = int(block)
        if partition is not None and (partition.block != self.block or self.gain != 1.0):
            partition = None
        self.partition = part = partition if partition is not None else partition_uniform(fir, self.block, self.gain)
        self.max_steps = int(max_steps)
        cfg = _capi.AdspUpolsConfig(self.device, self.chunk_size, self.channels, self.block, part.n_partitions, part.delay, self._fmt_code,
                                    self.max_steps)
        spectra = np.ascontiguousarray(part.spectra, dtype=np.float32)
        _capi.check(self._lib.adsp_upols_create(ctypes.byref(cfg), _ptr(spectra), ctypes.byref(self._h)))
        hist, blocks, nbytes = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_size_t(0)
        _capi.check(self._lib.adsp_upols_info(self._h, ctypes.byref(hist), ctypes.byref(blocks), ctypes.byref(nbytes)))
        self.history_chunks, self.delay_line_blocks, self.delay_line_bytes = hist.value, blocks.value, nbytes.value
        self.epilogue = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.adsp_upols_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _capi.check(self._lib.adsp_upols_reset(self._h))

    def set_epilogue(self, effect=None):
        """An effect (effects.Effect) applied to the output registers of the inverse transform; a tremolo's LFO starts at table index 0
        here and follows the stream (every channel in step, the reference's buffer quirk included)."""
        op, (p0, p1, p2) = (effect.op, effect.params()) if effect is not None else (_capi.EFFECT_NONE, (0.0, 0.0, 0.0))
        _capi.check(self._lib.adsp_upols_set_epilogue(self._h, int(op), float(p0), float(p1), float(p2)))
        self.epilogue = effect

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """Device-resident [n_steps, C, N] batches of the engine's sample type, not aliased; asynchronous on `stream`."""
        _capi.check(self._lib.adsp_upols_apply_device(self._h, _ptr(d_in), _ptr(d_out), int(n_steps), _ptr(stream)))

    def apply_host(self, x):
        if self.sample_format != "f32" and np.asarray(x).dtype != np.int16:
            raise TypeError("this engine filters int16 PCM; pass an int16 array")
        x = np.ascontiguousarray(x, dtype=self.dtype)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.channels, self.chunk_size):
            raise ValueError(f"expected [steps, {self.channels}, {self.chunk_size}], got {x.shape}")
        out = np.empty_like(x)
        _capi.check(self._lib.adsp_upols_apply_host(self._h, _ptr(x), _ptr(out), x.shape[0]))
        return out[0] if squeeze else out

    # -- the filter (dist.py: the path's one collective) ------------------------------------------------------------------------------
    @property
    def spectra_floats(self):
        return self.partition.n_partitions * 2 * (self.block + 1)

    def set_spectra(self, spectra_f32):
        """Replace the filter by partition spectra of the same shape ([n_partitions][block + 1] interleaved float32), e.g. received by broadcast."""
        sp = np.ascontiguousarray(spectra_f32, dtype=np.float32).reshape(-1)
        if sp.size != self.spectra_floats:
            raise ValueError(f"expected {self.spectra_floats} floats ({self.partition.n_partitions} partitions x {self.block + 1} bins), got {sp.size}")
        _capi.check(self._lib.adsp_upols_set_spectra(self._h, _ptr(sp)))

    def get_spectra(self):
        """The interleaved float32 partition spectra the engine's tables were last built from (after a broadcast: what the collective left)."""
        out = np.empty(self.spectra_floats, np.float32)
        _capi.check(self._lib.adsp_upols_get_spectra(self._h, _ptr(out), out.size))
        return out

    get_spectrum = get_spectra  # (the name bench.py's checksum and dist.py use for every engine kind)

    def bcast_rank(self, unique_id, rank, world, root=0):
        """adsp_upols_bcast_spectra_rank: one process per GPU; afterwards every rank's engine runs the root's filter."""
        uid = bytes(unique_id)
        if len(uid) != _capi.ADSP_RCCL_UNIQUE_ID_BYTES:
            raise ValueError(f"unique_id must be {_capi.ADSP_RCCL_UNIQUE_ID_BYTES} bytes")
        _capi.check(self._lib.adsp_upols_bcast_spectra_rank(self._h, uid, int(rank), int(world), int(root)))

    def synchronize(self, stream=None):
        """Wait for everything the engine has launched (on `stream`, or wherever its last call went)."""
        _capi.check(self._lib.adsp_upols_synchronize(self._h, _ptr(stream)))

    def set_carry(self, mode=-1):
        """adsp_upols_set_carry: the block that straddles the end of a call computed once and carried to the next call's output (1), in
        both calls (0), or as the library sees fit per call (-1, the default: carry from two workgroups per CU on).  Same samples."""
        _capi.check(self._lib.adsp_upols_set_carry(self._h, int(mode)))

    def get_state(self):
        """The engine's whole state as one uint8 array (counters, input ring, frequency-domain delay line: adsp_upols_get_state);
        set_state on an engine of the same configuration continues the stream bit for bit."""
        n = ctypes.c_size_t(0)
        _capi.check(self._lib.adsp_upols_state_bytes(self._h, ctypes.byref(n)))
        out = np.empty(n.value, np.uint8)
        _capi.check(self._lib.adsp_upols_get_state(self._h, _ptr(out), out.size))
        return out

    def set_state(self, state):
        st = np.ascontiguousarray(state, dtype=np.uint8)
        _capi.check(self._lib.adsp_upols_set_state(self._h, _ptr(st), st.size))


class MixBus:
    """MixSignals over filtered channels in one pass each (Utility.py:51-72 after K FFT devices): engine 0 writes the
    output buffer, the others add to it, the last one clips the sum - no separate mixing pass over HBM."""

    def __init__(self, engines):
        self.engines = list(engines)
        if len(self.engines) < 2:
            raise ValueError("a mix bus needs at least two engines")
        for i, eng in enumerate(self.engines):
            eng.set_accumulate(0 if i == 0 else (2 if i == len(self.engines) - 1 else 1))

    def apply_device(self, d_inputs, d_out, n_steps=1, stream=None):
        """d_inputs[k]: the [n_steps, C, N] device batch of engine k; all launches are ordered on `stream`."""
        for eng, d_in in zip(self.engines, d_inputs):
            eng.apply_device(d_in, d_out, n_steps, stream)


class Pcm16Adapter:
    """int16 PCM batches through a float32 engine whose kernels cannot take them directly - chunk sizes that are not multiples of 4 (the
    dword-access kernels are float32 only), kernels longer than one transform on such chunk sizes.  The arithmetic is the int16 engines':
    (float)x in, (int16)trunc(y) out with the reference's /32768 and *32767 folded into the taps (Utility.py:236-237, :306 -
    design.PCM16_GAIN); only the two conversions run as passes of their own (numpy on the host path, torch - plumbing - on the device path)."""

    def __init__(self, fir: FirStream, **kw):
        self.fir = fir
        scaled = FirStream(np.asarray(fir.taps, dtype=np.float64) * PCM16_GAIN, fir.chunk_size, fir.latency_chunks, fir.lookahead)
        self.inner = make_engine(scaled, **kw)
        self.channels, self.chunk_size, self.device = self.inner.channels, self.inner.chunk_size, self.inner.device
        self.sample_format, self.dtype, self.gain = "s16", np.int16, PCM16_GAIN

    def close(self):
        self.inner.close()

    def reset(self):
        self.inner.reset()

    def synchronize(self, stream=None):
        self.inner.synchronize(stream)

    def apply_host(self, x):
        if np.asarray(x).dtype != np.int16:
            raise TypeError("this engine filters int16 PCM; pass an int16 array")
        y = self.inner.apply_host(np.asarray(x).astype(np.float32))
        return np.trunc(y).astype(np.int32).astype(np.int16)  # truncation toward zero, low 16 bits kept: (x * 32767).astype('int16')

    def apply_device(self, d_in, d_out, n_steps=1, stream=None):
        """d_in / d_out: int16 torch tensors [n_steps, C, N] on the engine's GPU (raw addresses cannot be converted here)."""
        import torch
        if not (hasattr(d_in, "dtype") and d_in.dtype == torch.int16 and d_out.dtype == torch.int16):
            raise TypeError("the int16 adapter's device path takes torch.int16 tensors")
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream)) if isinstance(stream, int) and stream else torch.cuda.stream(torch.cuda.current_stream(d_in.device))
        with ctx:
            xf = d_in.to(torch.float32)
            yf = torch.empty_like(xf)
            self.inner.apply_device(xf, yf, n_steps, torch.cuda.current_stream(d_in.device).cuda_stream)
            d_out.copy_(torch.trunc(yf).to(torch.int32).to(torch.int16))


def make_engine(fir: FirStream, **kw):
    """FirEngine when the kernel fits one transform; otherwise the uniformly partitioned engine (UpolsFirEngine: one forward transform
    per input block, a frequency-domain delay line, one inverse per output block), and where its conditions do not hold (chunk sizes
    that are not multiples of 4, streams delayed by less than a block) PartitionedFirEngine (one engine pass per kernel slice).
    Keyword arguments an engine kind does not take (ring_slots / fft_mult beyond FirEngine, max_taps beyond PartitionedFirEngine) are
    dropped for the others."""
    n = int(fir.chunk_size)
    if kw.get("sample_format", "f32") == "s16" and (n % 4 != 0 or n < 16):
        # int16 PCM on a chunk size the 8-byte int16 accesses cannot tile: the float32 dword-access kernels behind two conversion passes
        return Pcm16Adapter(fir, **{k: v for k, v in kw.items() if k != "sample_format"})
    if fits_one_transform(fir):
        kw.pop("max_taps", None)
        kw.pop("max_steps", None)
        kw.pop("block", None)
        return FirEngine(fir, **kw)
    kw.pop("ring_slots", None)
    kw.pop("fft_mult", None)
    max_taps = kw.pop("max_taps", None)
    if upols_supported(fir):  # (the cheap preconditions: the partitions themselves - an rfft of the whole kernel - are computed once, by the engine)
        return UpolsFirEngine(fir, **kw)
    if kw.get("sample_format", "f32") != "f32":
        raise ValueError("kernels longer than one transform whose stream is delayed by less than a block are supported for float32 samples only")
    for k in ("sample_format", "max_steps", "block"):
        kw.pop(k, None)
    if max_taps is not None:
        kw["max_taps"] = max_taps
    return PartitionedFirEngine(fir, **kw)


def upols_supported(fir: FirStream, block=None):
    """The preconditions of the uniformly partitioned engine (adsp_upols_create), without designing anything: a chunk size that is a
    multiple of 4 (>= 16) and a stream delayed by at least one block."""
    block = int(block or _capi.load().adsp_upols_block_size())
    n = int(fir.chunk_size)
    return n % 4 == 0 and n >= 16 and int(fir.delay) - (int(fir.delay) % 4) >= block
