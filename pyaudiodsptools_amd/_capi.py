"""ctypes binding of libadsp.so (include/adsp.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADSP_LIB") or os.path.join(_HERE, "libadsp.so")  # ADSP_LIB: tuning builds only

ADSP_ABI_VERSION = 11
ADSP_MAX_HISTORY = 8
ADSP_RCCL_UNIQUE_ID_BYTES = 128
ADSP_FORMAT_F32, ADSP_FORMAT_S16, ADSP_FORMAT_S16_F64 = 0, 1, 2
EFFECT_NONE, EFFECT_VOLUME, EFFECT_SOFT_CLIPPER, EFFECT_HARD_DISTORTION, EFFECT_SATURATOR, EFFECT_TREMOLO = 0, 1, 2, 3, 4, 5
EFFECT_BIT_CRUSHER = 6
ADSP_OK, ADSP_ERR_ARG, ADSP_ERR_HIP, ADSP_ERR_STATE, ADSP_ERR_NO_DEVICE = 0, -1, -2, -3, -4


class AdspError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libadsp error {code}: {message}")
        self.code = code


class AdspConfig(ctypes.Structure):
    _fields_ = [
        ("device_id", ctypes.c_int),
        ("chunk_size", ctypes.c_int),
        ("n_channels", ctypes.c_int),
        ("fft_size", ctypes.c_int),
        ("history_chunks", ctypes.c_int),
        ("lookback", ctypes.c_int),
        ("out_offset", ctypes.c_int),
        ("ring_slots", ctypes.c_int),
        ("sample_format", ctypes.c_int),
    ]


class AdspDelayConfig(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("chunk_size", ctypes.c_int), ("n_channels", ctypes.c_int),
                ("n_taps", ctypes.c_int)]


class AdspScanConfig(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("chunk_size", ctypes.c_int), ("n_channels", ctypes.c_int),
                ("kind", ctypes.c_int), ("n_sections", ctypes.c_int)]


class AdspExactConfig(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("chunk_size", ctypes.c_int), ("n_channels", ctypes.c_int),
                ("n_taps", ctypes.c_int), ("delay", ctypes.c_int), ("sample_format", ctypes.c_int)]


class AdspUpolsConfig(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("chunk_size", ctypes.c_int), ("n_channels", ctypes.c_int), ("block_size", ctypes.c_int),
                ("n_partitions", ctypes.c_int), ("delay", ctypes.c_int), ("sample_format", ctypes.c_int), ("max_steps", ctypes.c_int)]


_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_float_p = ctypes.POINTER(ctypes.c_float)
_engine_p = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/adsp.h declares
SIGNATURES = {
    "adsp_version": (ctypes.c_int, []),
    "adsp_last_error": (ctypes.c_char_p, []),
    "adsp_build_info": (ctypes.c_char_p, []),
    "adsp_device_count": (ctypes.c_int, [_c_int_p]),
    "adsp_plan_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "adsp_plan_describe": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _c_int_p, _c_int_p, _c_int_p, _c_int_p, _c_int_p]),
    "adsp_create": (ctypes.c_int, [ctypes.POINTER(AdspConfig), ctypes.POINTER(_engine_p)]),
    "adsp_destroy": (ctypes.c_int, [_engine_p]),
    "adsp_set_spectrum": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_set_spectrum_f64": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_set_spectrum_async": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_set_spectrum_device": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_set_kernel_reach": (ctypes.c_int, [_engine_p, ctypes.c_int]),
    "adsp_set_block_outputs": (ctypes.c_int, [_engine_p, ctypes.c_int]),
    "adsp_spectrum_is_real": (ctypes.c_int, [_engine_p, _c_int_p]),
    "adsp_set_epilogue": (ctypes.c_int, [_engine_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float]),
    "adsp_effect_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "adsp_effect_host": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "adsp_tremolo_rows_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "adsp_mix_device": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "adsp_mix_host": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_size_t]),
    "adsp_nonfinite_guard": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p]),
    "adsp_get_epilogue_state": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_longlong)]),
    "adsp_set_epilogue_state": (ctypes.c_int, [_engine_p, ctypes.c_longlong]),
    "adsp_set_accumulate": (ctypes.c_int, [_engine_p, ctypes.c_int]),
    "adsp_scan_create_biquad": (ctypes.c_int, [ctypes.POINTER(AdspScanConfig), ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_scan_create_compressor": (ctypes.c_int, [ctypes.POINTER(AdspScanConfig), ctypes.c_float, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_scan_create_gate": (ctypes.c_int, [ctypes.POINTER(AdspScanConfig), ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_scan_destroy": (None, [ctypes.c_void_p]),
    "adsp_scan_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "adsp_scan_apply_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_scan_apply_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_delay_create": (ctypes.c_int, [ctypes.POINTER(AdspDelayConfig), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                         ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_delay_destroy": (None, [ctypes.c_void_p]),
    "adsp_delay_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "adsp_delay_set_accumulate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "adsp_delay_history_chunks": (ctypes.c_int, [ctypes.c_void_p, _c_int_p]),
    "adsp_delay_apply_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_delay_apply_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_exact_create": (ctypes.c_int, [ctypes.POINTER(AdspExactConfig), ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_exact_destroy": (None, [ctypes.c_void_p]),
    "adsp_exact_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "adsp_exact_apply_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_exact_apply_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_upols_block_size": (ctypes.c_int, []),
    "adsp_upols_block_sizes": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "adsp_upols_create": (ctypes.c_int, [ctypes.POINTER(AdspUpolsConfig), ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_upols_destroy": (None, [ctypes.c_void_p]),
    "adsp_upols_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "adsp_upols_set_epilogue": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float]),
    "adsp_upols_info": (ctypes.c_int, [ctypes.c_void_p, _c_int_p, _c_int_p, ctypes.POINTER(ctypes.c_size_t)]),
    "adsp_upols_apply_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_upols_apply_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_upols_set_spectra": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "adsp_upols_get_spectra": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "adsp_upols_bcast_spectra": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    "adsp_upols_bcast_spectra_rank": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "adsp_upols_synchronize": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "adsp_upols_set_carry": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "adsp_upols_state_bytes": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]),
    "adsp_upols_get_state": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "adsp_upols_set_state": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "adsp_reset": (ctypes.c_int, [_engine_p]),
    "adsp_apply_host": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_apply_device": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_ring_acquire": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_ring_acquire_stream": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "adsp_ring_reset_order": (ctypes.c_int, [_engine_p]),
    "adsp_ring_set_pipeline": (ctypes.c_int, [_engine_p, ctypes.c_int]),
    "adsp_ring_join": (ctypes.c_int, [_engine_p, ctypes.c_void_p]),
    "adsp_ring_produce_begin": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "adsp_ring_produce_end": (ctypes.c_int, [_engine_p, ctypes.c_void_p]),
    "adsp_apply_ring_resident": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "adsp_ring_resident_timeout": (ctypes.c_int, [_engine_p, ctypes.c_double]),
    "adsp_ring_resident_status": (ctypes.c_int, [_engine_p, _c_int_p]),
    "adsp_live_configure": (ctypes.c_int, [_engine_p, ctypes.c_double, ctypes.c_int]),
    "adsp_live_start": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]),
    "adsp_live_slot": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_live_publish_host": (ctypes.c_int, [_engine_p]),
    "adsp_live_publish_stream": (ctypes.c_int, [_engine_p, ctypes.c_void_p]),
    "adsp_live_publish_run": (ctypes.c_int, [_engine_p, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]),
    "adsp_live_progress": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_uint)]),
    "adsp_live_wait": (ctypes.c_int, [_engine_p, ctypes.c_uint, ctypes.c_double]),
    "adsp_live_device_words": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_live_stop": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_uint)]),
    "adsp_bcast_spectrum": (ctypes.c_int, [ctypes.POINTER(_engine_p), ctypes.c_int, ctypes.c_int]),
    "adsp_rccl_version": (ctypes.c_int, [_c_int_p]),
    "adsp_rccl_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "adsp_bcast_spectrum_rank": (ctypes.c_int, [_engine_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "adsp_rccl_finalize": (ctypes.c_int, []),
    "adsp_get_spectrum": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_int]),
    "adsp_clock_probe_launch": (ctypes.c_int, [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "adsp_clock_probe_read": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "adsp_apply_ring": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_void_p]),
    "adsp_get_state": (ctypes.c_int, [_engine_p, ctypes.c_void_p]),
    "adsp_set_state": (ctypes.c_int, [_engine_p, ctypes.c_void_p]),
    "adsp_enable_kernel_timing": (ctypes.c_int, [_engine_p, ctypes.c_int]),
    "adsp_kernel_time": (ctypes.c_int, [_engine_p, ctypes.POINTER(ctypes.c_double), _c_int_p]),
    "adsp_synchronize": (ctypes.c_int, [_engine_p, ctypes.c_void_p]),
    "adsp_synth_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
}

_lib = None


def load():
    """dlopen libadsp.so (once) and declare every prototype.  Raises ImportError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C pyaudiodsptools_amd/csrc`. "
            "pyaudiodsptools_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.
    # If libadsp pulled in /opt/rocm's copy first, torch.cuda would later find "no HIP GPUs".  Loading
    # torch's runtime first makes libadsp's DT_NEEDED resolve (by SONAME) to the same copy, so torch
    # tensors, streams and this library share one device context.  torch is optional.
    try:
        import torch  # noqa: F401
        # ... and one RCCL: adsp_bcast_spectrum opens librccl.so on first use; point it at the copy torch bundles (built
        # against the HIP runtime that is now loaded) unless the caller chose one
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("ADSP_RCCL_LIB", rccl)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        except AttributeError:
            if not os.environ.get("ADSP_LIB"):
                raise
            continue  # a tuning library (ADSP_LIB) built before this entry point existed: A/B runs do not call it
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.adsp_version() != ADSP_ABI_VERSION:
        raise ImportError(f"libadsp ABI {lib.adsp_version()} != binding {ADSP_ABI_VERSION}")
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().adsp_last_error()
        raise AdspError(code, msg.decode("utf-8", "replace") if msg else "")


def device_count():
    n = ctypes.c_int(0)
    rc = load().adsp_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def plan_describe(chunk_size, fft_size):
    vals = [ctypes.c_int(0) for _ in range(5)]
    check(load().adsp_plan_describe(chunk_size, fft_size, *[ctypes.byref(v) for v in vals]))
    keys = ("complex_points", "points_per_thread", "threads_per_transform", "channels_per_workgroup", "lds_bytes")
    return dict(zip(keys, (v.value for v in vals)))
