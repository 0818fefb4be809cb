"""CPU-side checks: the C-ABI library loads, exports every symbol include/adsp.h declares, plans and
geometry are consistent, and compute entry points fail loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from pyaudiodsptools_amd import _capi
    lib = _capi.load()
    header = open(os.path.join(ROOT, "include", "adsp.h")).read()
    declared = sorted(set(re.findall(r"ADSP_API\s+[\w\s\*]+?\b(adsp_\w+)\s*\(", header)))
    assert len(declared) >= 18
    assert set(declared) == set(_capi.SIGNATURES), (declared, sorted(_capi.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.adsp_version() == _capi.ADSP_ABI_VERSION


def test_plans_cover_supported_chunk_sizes():
    from pyaudiodsptools_amd import _capi
    lib = _capi.load()
    for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        for f in (2 * n, 4 * n):
            assert lib.adsp_plan_supported(n, f) == 0, (n, f)
            d = _capi.plan_describe(n, f)
            assert d["complex_points"] == f // 2
            assert d["threads_per_transform"] * d["points_per_thread"] == f // 2
            assert d["lds_bytes"] == d["channels_per_workgroup"] * (f // 2) * 8 <= 160 * 1024
            assert (n // 4) % (2 * d["threads_per_transform"]) == 0  # design.py's N/4 granularity is legal
    assert lib.adsp_plan_supported(3000, 6000) != 0   # transforms are powers of two
    assert lib.adsp_plan_supported(32, 64) != 0       # ... of at least 128 points
    assert lib.adsp_plan_supported(3002, 8192) != 0   # chunk must be a multiple of 4
    assert lib.adsp_plan_supported(3000, 8192) == 0   # generic geometry: any such chunk with any supported transform
    assert lib.adsp_plan_supported(44100, 32768) == 0
    assert b"chunk_size" in lib.adsp_last_error() or b"fft_size" in lib.adsp_last_error()


def test_geometry_matches_header_documentation():
    from pyaudiodsptools_amd import design
    for n in (64, 512, 4096, 8192):
        lc = design.FirStream(design.lowcut_kernel(800, 44100, n), n)
        g = design.overlap_save_geometry(lc)
        # symmetric kernel centred on circular index 0 (real spectrum): the kept slice starts after d = N/4 - 1 wrapped taps
        assert (g.fft_size, g.history_chunks, g.lookback, g.out_offset, g.shift, g.max_block_outputs, g.zero_phase) == \
            (2 * n, 2, n + n // 4, n // 4, -(n // 4 - 1), n + n // 2, True)
        spec = design.engine_spectrum(lc, g)
        assert np.all(spec[1::2] == 0) and np.abs(spec[0::2]).max() > 0.5
        eq = design.FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, 44100, n), n)
        g = design.overlap_save_geometry(eq)
        assert (g.fft_size, g.history_chunks, g.lookback, g.out_offset, g.shift, g.max_block_outputs, g.zero_phase) == \
            (2 * n, 2, 2 * n - n // 4, n, 1, n, False)  # not symmetric: the mid band sits d samples later than the shelves
        hc = design.FirStream(design.highcut_kernel(8000, 44100, n), n)
        ch = lc.then(eq).then(hc)
        g = design.overlap_save_geometry(ch)
        assert len(ch.taps) == 4 * (n // 2 - 1) - 3 and ch.latency_chunks == 3
        assert (g.fft_size, g.history_chunks, g.out_offset) == (4 * n, 5, 2 * n)


def test_design_matches_reference_kernels(golden):
    from pyaudiodsptools_amd import design
    g = golden["design"]
    assert np.abs(design.lowcut_kernel(800, 44100, 4096) - g["lowcut_44100_4096_800"]).max() < 1e-12
    assert np.abs(design.highcut_kernel(8000, 96000, 8192) - g["highcut_96000_8192_8000"]).max() < 1e-12
    k = design.eq3_kernels(250, 1500, 6000, 48000, 1024)
    for band in ("highshelf", "lowshelf", "mid_lowpass", "mid_highpass"):
        assert np.abs(k[band] - g[f"eq_48000_1024_250_-6_1500_3_6000_-2.5_{band}"]).max() < 1e-12
    spec = design.reference_spectrum_3n(design.highcut_kernel(8000, 44100, 4096), 4096)
    assert abs(spec[1] - g["spot_B_H01"][1]) < 1e-12


def test_host_overlap_save_math_against_golden(golden):
    """design.py's geometry + spectrum, executed with numpy's rfft instead of the GPU kernel, reproduces
    the reference streams: validates everything on the host side of the ABI without a GPU."""
    from pyaudiodsptools_amd import design
    from conftest import assert_parity, seeded_stream
    cases = {"A": (design.lowcut_kernel(800, 44100, 4096), 4096, 1234, 6),
             "C": (design.eq3_composite(100, 2, 700, -4, 8000, 5, 44100, 512), 512, 1234, 6),
             "HC256": (design.highcut_kernel(3000, 44100, 256), 256, 83, 9)}
    for name, (taps, n, seed, chunks) in cases.items():
        fir = design.FirStream(taps, n)
        geo = design.overlap_save_geometry(fir)
        spec = design.engine_spectrum(fir, geo).view(np.complex64).astype(np.complex128)
        x = seeded_stream(seed, chunks * n).astype(np.float64)
        padded = np.concatenate([np.zeros(geo.history_chunks * n), x, np.zeros(geo.fft_size)])
        for v in (n, geo.max_block_outputs):
            out = np.zeros(chunks * n + v)
            for o in range(0, chunks * n, v):
                a = o - geo.lookback + geo.history_chunks * n
                y = np.fft.irfft(np.fft.rfft(padded[a:a + geo.fft_size]) * spec, geo.fft_size)
                out[o:o + v] = y[geo.out_offset:geo.out_offset + v]
            assert_parity(out[:chunks * n], golden["kat_streams"][name], what=f"{name} V={v}")


def test_no_gpu_means_loud_failure():
    from pyaudiodsptools_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    import pyaudiodsptools_amd as adsp
    adsp.config.initialize(44100, 512)
    with pytest.raises(_capi.AdspError) as ei:
        adsp.CreateLowCutFilter(800)
    assert ei.value.code == _capi.ADSP_ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    import ast
    pkg = os.path.join(ROOT, "pyaudiodsptools_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n.split(".")[0] == "oracle" for n in names), fn
    for fn in os.listdir(os.path.join(pkg, "csrc")):
        if fn.endswith((".hip", ".hpp", ".h", ".cpp")) or fn == "Makefile":
            assert "oracle" not in open(os.path.join(pkg, "csrc", fn)).read(), fn


def test_config_mirror():
    from pyaudiodsptools_amd import config
    config.initialize(48000, 1024)
    assert (config.sampling_rate, config.chunk_size, config.use_gpu) == (48000, 1024, False)
    config.initialize(44100, 512, use_gpu=True)
    assert (config.sampling_rate, config.chunk_size, config.use_gpu) == (44100, 512, True)
