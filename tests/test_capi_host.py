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
            full = d["channels_per_workgroup"] * (f // 2) * 8  # one complex64 per point; the large transforms exchange through half a buffer
            two_wave_4096 = f // 2 == 4096 and f == 4 * n  # round 5: the 4N transform of N = 2048 runs on 32 points per thread in two waves,
            one_wave_2048 = f // 2 == 2048 and f == 4 * n  # that of N = 1024 in one: both exchange through half a buffer
            assert d["lds_bytes"] == (full // 2 if f // 2 >= 8192 or two_wave_4096 or one_wave_2048 else full) <= 64 * 1024
            if f // 2 == 4096:
                assert (d["points_per_thread"], d["threads_per_transform"]) == ((32, 128) if two_wave_4096 else (16, 256))
            assert (n // 4) % (2 * d["threads_per_transform"]) == 0  # design.py's N/4 granularity is legal
    assert lib.adsp_plan_supported(3000, 6000) != 0   # transforms are powers of two ...
    assert lib.adsp_plan_supported(4096, 6144) == 0   # ... or 1.5 x a power-of-two chunk that has a 3 * 2^k plan (M = 3072)
    d = _capi.plan_describe(4096, 6144)
    assert (d["complex_points"], d["points_per_thread"], d["threads_per_transform"], d["lds_bytes"]) == (3072, 48, 64, 3072 * 8 // 2)
    assert lib.adsp_plan_supported(512, 768) != 0 and lib.adsp_plan_supported(1000, 6144) != 0
    assert lib.adsp_plan_supported(32, 64) != 0       # ... of at least 128 points
    assert lib.adsp_plan_supported(3002, 8192) == 0   # round 4: ANY chunk size >= 4 (not a multiple of 4: dword-access kernel)
    assert lib.adsp_plan_supported(1001, 4096) == 0 and lib.adsp_plan_supported(30, 128) == 0 and lib.adsp_plan_supported(6, 128) == 0
    assert lib.adsp_plan_supported(3, 128) != 0
    assert lib.adsp_plan_supported(3000, 8192) == 0   # generic geometry: any such chunk with any supported transform
    assert lib.adsp_plan_supported(44100, 32768) == 0
    assert b"chunk_size" in lib.adsp_last_error() or b"fft_size" in lib.adsp_last_error()


def test_geometry_matches_header_documentation():
    from pyaudiodsptools_amd import design
    for n in (64, 512, 4096, 8192):
        lc = design.FirStream(design.lowcut_kernel(800, 44100, n), n)
        g = design.overlap_save_geometry(lc, 2)
        # symmetric kernel centred on circular index 0 (real spectrum): the kept slice starts after d = N/4 - 1 wrapped taps
        assert (g.fft_size, g.history_chunks, g.lookback, g.out_offset, g.shift, g.max_block_outputs, g.zero_phase) == \
            (2 * n, 2, n + n // 4, n // 4, -(n // 4 - 1), n + n // 2, True)
        gb = design.overlap_save_geometry(lc, 0, "batch")  # multi-step launches: 1.5 N kept per 2N transform, or, where the
        if n == 4096:                                         # 4N transform runs on a fast plan (N = 1024..4096), 3.5 N of 4 N
            assert (gb.fft_size, gb.history_chunks, gb.lookback, gb.out_offset, gb.shift, gb.max_block_outputs, gb.zero_phase) == \
                (4 * n, 2, n + n // 4, n // 4, -(n // 4 - 1), 3 * n + n // 2, True)
        else:
            assert gb == g
        assert design.overlap_save_geometry(lc) == g
        if n == 4096:  # opt-in: the minimal window N + 2d = 1.5 N of single-step launches (the 3 * 2^k plan, M = 3072)
            gs = design.overlap_save_geometry(lc, 1.5)
            assert (gs.fft_size, gs.history_chunks, gs.lookback, gs.out_offset, gs.shift, gs.max_block_outputs, gs.zero_phase) == \
                (3 * n // 2, 2, n + n // 4, n // 4, -(n // 4 - 1), n, True)
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
    # even filter lengths (N // 2 odd: N = 1002 -> 500 taps, N = 30 -> 14)
    assert np.abs(design.lowcut_kernel(500, 48000, 1002) - g["lowcut_48000_1002_500"]).max() < 1e-12
    assert np.abs(design.highcut_kernel(500, 48000, 1002) - g["highcut_48000_1002_500"]).max() < 1e-12
    assert np.abs(design.lowcut_kernel(3000, 44100, 30) - g["lowcut_44100_30_3000"]).max() < 1e-12
    k = design.eq3_kernels(250, 1500, 6000, 48000, 1002)
    for band in ("highshelf", "lowshelf", "mid_lowpass", "mid_highpass"):
        assert np.abs(k[band] - g[f"eq_48000_1002_250_-6_1500_3_6000_-2.5_{band}"]).max() < 1e-12
    assert design.filter_length(1002) == (500, 250) and design.filter_length(4096) == (2047, 1023) and design.filter_length(30) == (14, 7)
    spec = design.reference_spectrum_3n(design.highcut_kernel(8000, 44100, 4096), 4096)
    assert abs(spec[1] - g["spot_B_H01"][1]) < 1e-12


def test_host_overlap_save_math_against_golden(golden):
    """design.py's geometry + spectrum, executed with numpy's rfft instead of the GPU kernel, reproduces
    the reference streams: validates everything on the host side of the ABI without a GPU."""
    from pyaudiodsptools_amd import design
    from conftest import assert_parity, seeded_stream
    cases = {"A": (design.lowcut_kernel(800, 44100, 4096), 4096, 1234, 6),
             "C": (design.eq3_composite(100, 2, 700, -4, 8000, 5, 44100, 512), 512, 1234, 6),
             "HC256": (design.highcut_kernel(3000, 44100, 256), 256, 83, 9),
             # chunk sizes that are not multiples of 4: even filter lengths (look-ahead L // 2), generic geometry with nothing aligned
             "LC30": (design.lowcut_kernel(3000, 44100, 30), 30, 93, 30),
             "EQ30": (design.eq3_composite(100, 2, 700, -4, 8000, 5, 44100, 30), 30, 95, 30),
             "LC1001": (design.lowcut_kernel(300, 44100, 1001), 1001, 96, 7),
             "HC1002": (design.highcut_kernel(9000, 48000, 1002), 1002, 99, 7),
             "EQ1002": (design.eq3_composite(250, -6, 1500, 3, 6000, -2.5, 48000, 1002), 1002, 100, 7),
             "HC6": (design.highcut_kernel(8000, 44100, 6), 6, 101, 50),
             "LC4410": (design.lowcut_kernel(160, 44100, 4410), 4410, 102, 4)}
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


def test_trimmed_chain_on_the_host_side_of_the_abi(golden):
    """The fused LowCut -> EQ3 -> HighCut chain with its negligible end taps left out (FirStream.trimmed): geometry,
    error bound, and - with numpy's rfft in place of the kernel, window tail zeroed like the kernel's window skip - golden E."""
    from pyaudiodsptools_amd import design
    from conftest import assert_parity, seeded_stream
    n, fs = 8192, 96000
    lc = design.FirStream(design.lowcut_kernel(800, fs, n), n)
    eq = design.FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n), n)
    hc = design.FirStream(design.highcut_kernel(8000, fs, n), n)
    full = lc.then(eq).then(hc)
    fir = full.trimmed()
    dropped_front = full.lookahead - fir.lookahead
    assert len(full.taps) == 16377 and len(fir.taps) == 9401 and fir.delay == full.delay + dropped_front
    kept = np.zeros(len(full.taps), bool)
    kept[dropped_front:dropped_front + len(fir.taps)] = True
    assert np.array_equal(full.taps[kept], fir.taps)
    assert np.abs(full.taps[~kept]).sum() <= design.TRIM_EPS * np.abs(full.taps).sum()  # the worst-case output change
    assert full.trimmed(0.0) is full and lc.trimmed() is not None
    geo, geo_full = design.overlap_save_geometry(fir, 0, "batch"), design.overlap_save_geometry(full, 0, "batch")
    assert (geo.fft_size, geo.history_chunks, geo.max_block_outputs) == (4 * n, 4, 11 * n // 4)
    assert (geo_full.fft_size, geo_full.history_chunks, geo_full.max_block_outputs) == (4 * n, 5, 2 * n)
    spec = design.engine_spectrum(fir, geo).view(np.complex64).astype(np.complex128)
    chunks = 12
    x = seeded_stream(4321, chunks * n).astype(np.float64)
    padded = np.concatenate([np.zeros(geo.history_chunks * n), x, np.zeros(geo.fft_size)])
    for v in (n, geo.max_block_outputs):
        reach = max(0, -geo.shift)
        out = np.zeros(chunks * n + v)
        for o in range(0, chunks * n, v):
            a = o - geo.lookback + geo.history_chunks * n
            win = padded[a:a + geo.fft_size].copy()
            win[geo.out_offset + v + reach:] = 0.0  # what adsp_set_kernel_reach lets the kernel leave unfetched
            y = np.fft.irfft(np.fft.rfft(win) * spec, geo.fft_size)
            out[o:o + v] = y[geo.out_offset:geo.out_offset + v]
        assert_parity(out[:chunks * n], golden["kat_chain"]["E"], what=f"trimmed chain V={v}")


def test_window_tail_skip_is_exact_for_zero_phase_filters(golden):
    """adsp_set_kernel_reach on the host side: for a zero-phase cut filter a single-step block (V = N) only depends on the
    first out_offset + N + (L-1)/2 = 1.5 N - 1 window positions - zeroing the rest changes nothing."""
    from pyaudiodsptools_amd import design
    n = 512
    fir = design.FirStream(design.lowcut_kernel(200, 44100, n), n)
    geo = design.overlap_save_geometry(fir)
    assert geo.zero_phase and geo.shift == -(len(fir.taps) - 1) // 2
    spec = design.engine_spectrum(fir, geo).view(np.complex64).astype(np.complex128)
    win = np.random.default_rng(3).uniform(-1, 1, geo.fft_size)
    need = geo.out_offset + n - geo.shift
    assert need == 3 * n // 2 - 1
    cut = win.copy()
    cut[need:] = 0.0
    full = np.fft.irfft(np.fft.rfft(win) * spec, geo.fft_size)[geo.out_offset:geo.out_offset + n]
    part = np.fft.irfft(np.fft.rfft(cut) * spec, geo.fft_size)[geo.out_offset:geo.out_offset + n]
    # (exactly nothing in exact arithmetic; the float32 rounding of the spectrum spreads 1e-8 of the kernel over all F taps)
    assert np.abs(full - part).max() <= 1e-7
    cut[need - 96:] = 0.0  # noticeably fewer positions are NOT enough (the outermost taps of a Blackman window are ~0)
    less = np.fft.irfft(np.fft.rfft(cut) * spec, geo.fft_size)[geo.out_offset:geo.out_offset + n]
    assert np.abs(full - less).max() > 1e-5


def test_cpu_baseline_helpers():
    """bench.py's cpu_baseline leg (oracle/cpu_bench.py): worker returns samples and seconds, core detection is sane."""
    from oracle import cpu_bench
    assert 1 <= cpu_bench.physical_cores() <= (os.cpu_count() or 1)
    samples, el = cpu_bench.run_worker(("lowcut", "literal3n", 512, 44100, 1, 0.05, 1))
    assert samples > 0 and el >= 0.05
    samples, el = cpu_bench.run_worker(("eq3", "rfft2n", 512, 44100, 4, 0.05, 1))
    assert samples > 0 and samples % (4 * 512) == 0
    assert cpu_bench.run_worker(("chain", "rfft2n", 512, 44100, 4, 0.05, 1)) == (0, 0.0)  # does not fit a 2N transform


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


def test_geometry_invariants_and_host_overlap_save_for_random_kernels():
    """design.overlap_save_geometry for 150 random kernels (lengths, delays, symmetric or not, power-of-two and arbitrary
    chunk sizes, stream and batch): alignment rules of include/adsp.h hold, and the overlap-save arithmetic executed with
    numpy's rfft (window tail zeroed where the kernel would skip it) equals the float64 direct convolution."""
    from pyaudiodsptools_amd import _capi, design
    from oracle import fftfilter_oracle as o
    rng = np.random.default_rng(2024)
    done = 0
    while done < 150:
        n = int(rng.choice([64, 128, 256, 512, 1024, 2048, 100, 360, 1000, 1920]))
        m = int(rng.integers(1, n))
        sym = rng.random() < 0.4
        taps = rng.normal(size=m)
        if sym:
            taps = taps + taps[::-1]
        latency = int(rng.integers(1, 3))
        lookahead = int(rng.integers(0, n // 2))
        fir = design.FirStream(taps / max(np.abs(taps).sum(), 1e-9), n, latency, lookahead)
        opt = str(rng.choice(["stream", "batch"]))
        try:
            geo = design.overlap_save_geometry(fir, 0, opt)
        except ValueError:
            continue
        done += 1
        pow2 = n & (n - 1) == 0
        f, g = geo.fft_size, (n // 4 if pow2 else 4)
        reach = max(0, -geo.shift)
        assert f & (f - 1) == 0 and 128 <= f <= 32768
        assert geo.lookback % g == 0 and 0 < geo.lookback <= geo.history_chunks * n <= _capi.ADSP_MAX_HISTORY * n
        assert geo.out_offset % g == 0 and geo.out_offset <= geo.lookback
        assert geo.max_block_outputs >= (n if pow2 else 1) and geo.out_offset + geo.max_block_outputs <= f - reach
        # circular placement: no tap wraps into the kept slice
        assert geo.out_offset >= geo.shift + m - 1 if geo.shift >= 0 else geo.out_offset >= reach
        spec = design.engine_spectrum(fir, geo).view(np.complex64).astype(np.complex128)
        chunks = 5
        x = rng.uniform(-1, 1, chunks * n)
        truth = o.direct_stream_convolution(fir.taps, x, n, fir.latency_chunks, fir.lookahead)
        padded = np.concatenate([np.zeros(geo.history_chunks * n), x, np.zeros(f)])
        v = n if (opt == "stream" and pow2) else geo.max_block_outputs
        out = np.zeros(chunks * n + v)
        for ob in range(0, chunks * n, v):
            a = ob - geo.lookback + geo.history_chunks * n
            win = padded[a:a + f].copy()
            win[geo.out_offset + v + reach:] = 0.0
            out[ob:ob + v] = np.fft.irfft(np.fft.rfft(win) * spec, f)[geo.out_offset:geo.out_offset + v]
        err = np.abs(out[:chunks * n] - truth).max()
        assert err <= 2e-6 * max(np.abs(truth).max(), 1e-3) + 1e-7, (n, m, sym, latency, lookahead, opt, geo, err)


def _build_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "capi_demo")
    pkg = os.path.join(ROOT, "pyaudiodsptools_amd")
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "capi_demo.c"),
           "-L" + pkg, "-ladsp", "-lm", "-Wl,-rpath," + pkg, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return exe


def test_plain_c_program_links_against_the_abi_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/capi_demo.c: include/adsp.h is a C header (compiled as C11 with -Wall -Werror) and libadsp.so links from C.
    Without a GPU the program stops at adsp_create with libadsp's message (there is no CPU fallback)."""
    import subprocess
    from pyaudiodsptools_amd import _capi
    exe = _build_c_demo(tmp_path)
    if _capi.device_count() >= 1:
        pytest.skip("a GPU is visible: the run itself is tests/test_gpu_parity.py::test_plain_c_program_through_the_abi")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "adsp_create" in out.stderr and "no HIP device" in out.stderr


def test_traffic_stamp_follows_the_kernel_code_not_its_comments(tmp_path):
    """profiles/traffic.json entries are stamped with bench.kernel_sha16(); bench.py reports `traffic: null` when the stamp
    differs from the sources it runs on.  The identity must change with the code and must not change with a comment."""
    import json
    import shutil
    import bench
    src = os.path.join(ROOT, "pyaudiodsptools_amd", "csrc")
    files = ("fftconv_kernel.hpp", "fftconv_core.inc", "plan_table.hpp", "plan_table_core.inc")
    for f in files:
        shutil.copy(os.path.join(src, f), tmp_path / f)
    base = bench.kernel_sha16(str(tmp_path))
    assert base == bench.kernel_sha16() and len(base) == 16
    with open(tmp_path / "fftconv_core.inc", "a") as fh:
        fh.write("\n// a remark\n\n")
    assert bench.kernel_sha16(str(tmp_path)) == base
    with open(tmp_path / "fftconv_core.inc", "a") as fh:
        fh.write("static const int changed = 1;\n")
    assert bench.kernel_sha16(str(tmp_path)) != base
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    stamped = [k for k, v in traffic.items() if isinstance(v, dict) and "kernel_sha16" in v]
    assert "lowcut_4096x4096_batch" in stamped and all(len(traffic[k]["kernel_sha16"]) == 16 for k in stamped)
    # every stamped entry names the committed rocprofv3 summary it was folded from, and that file holds both traffic passes
    for k in stamped:
        summary = os.path.join(ROOT, traffic[k]["source"])
        assert os.path.isfile(summary), (k, traffic[k]["source"])
        txt = open(summary).read()
        assert "FETCH_SIZE" in txt and "WRITE_SIZE" in txt, summary
    if traffic["lowcut_4096x4096_batch"]["kernel_sha16"] != base:  # not an error here: bench.py then reports traffic: null and says why
        import warnings
        warnings.warn("profiles/traffic.json was measured on other kernel sources than this tree's: re-run tools/sessions/r4_session25.sh "
                      "and tools/update_traffic.py before quoting a traffic figure")


def test_upols_block_sizes_are_reported_without_a_gpu():
    """adsp_upols_block_sizes: count with a NULL array, the ascending sizes otherwise; the smallest is adsp_upols_block_size()."""
    from pyaudiodsptools_amd import _capi
    lib = _capi.load()
    assert lib.adsp_upols_block_sizes(None, 0) == 2
    sizes = (ctypes.c_int * 4)()
    assert lib.adsp_upols_block_sizes(sizes, 4) == 2 and list(sizes[:2]) == [8192, 16384] and lib.adsp_upols_block_size() == 8192
    one = (ctypes.c_int * 1)()
    assert lib.adsp_upols_block_sizes(one, 1) == 2 and one[0] == 8192   # a short array is filled as far as it goes


def test_product_library_has_no_tuning_scaffolding():
    """the A/B plan variants, the ablation switches and the persistent-block kernels live in libadsp_tuning.so
    (`make tuning`); the product library says what it is, refuses ADSP_PLAN_VARIANT with a message, and its Makefile rule takes no
    EXTRA flags (a stray EXTRA=-DADSP_ABLATE=... cannot change what libadsp.so computes; the macros do not even compile there)."""
    import subprocess
    import sys
    from pyaudiodsptools_amd import _capi
    lib = _capi.load()
    assert lib.adsp_build_info() == b"product"
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['ADSP_PLAN_VARIANT'] = '22'\n"
            "from pyaudiodsptools_amd import _capi\n"
            "lib = _capi.load(); rc = lib.adsp_plan_supported(4096, 8192); print(rc, lib.adsp_last_error().decode())\n"
            "os.environ['ADSP_PLAN_VARIANT'] = ''; print(lib.adsp_plan_supported(4096, 8192))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0].startswith("-3 ") and "libadsp_tuning.so" in out[0] and "make" in out[0], out
    assert out[1].strip() == "0", "an EMPTY ADSP_PLAN_VARIANT is 'not set'"
    symbols = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    mk = open(os.path.join(ROOT, "pyaudiodsptools_amd", "csrc", "Makefile")).read()
    product_rule = mk[mk.index("%.o: %.hip"):mk.index("tuning:")]
    assert "$(EXTRA)" not in product_rule and "plans_var" not in mk[mk.index("SRC"):mk.index("OBJ")]
    hdr = open(os.path.join(ROOT, "pyaudiodsptools_amd", "csrc", "fftconv_kernel.hpp")).read()
    assert "#error" in hdr and "ADSP_TUNING_BUILD" in hdr
    assert symbols.count(" T adsp_") == len(_capi.SIGNATURES)
