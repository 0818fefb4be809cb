"""The small DFTs of csrc/fftconv_core.inc (the butterflies of every Stockham pass) compiled for the HOST and checked against a direct
float64 DFT: the lines between the [dft-begin] / [dft-end] markers are plain C++ once `__device__` is defined away.  Covers both forms of
the radix-2 combine (ADSP_FUSED_BFLY = 1: six multiply-adds with tan / cot constants; 0: the separate twiddle product) and both entry
points (run: natural-order input; run_pairs: the caller has taken the first radix-2 level - where Pass::compute folds the pass twiddles in)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORE = os.path.join(ROOT, "pyaudiodsptools_amd", "csrc", "fftconv_core.inc")

HARNESS = r"""
#include <cstdio>
#include <cstdlib>
#include <utility>
#define __device__
#define __host__
#define __forceinline__ inline
#define ADSP_F64 %(f64)d
#if ADSP_F64
using real = double;
#define ADSP_LIT(x) x
#else
using real = float;
#define ADSP_LIT(x) x##f
#endif
#define ADSP_FUSED_BFLY %(fused)d
%(section)s

template <int R>
void one(const double* in) {
    real xr[R], xi[R], yr[R], yi[R];
    for (int q = 0; q < R; ++q) { xr[q] = (real)in[2 * q]; xi[q] = (real)in[2 * q + 1]; }
    Dft<R>::run(xr, xi, yr, yi);
    printf("run %%d", R);
    for (int q = 0; q < R; ++q) printf(" %%.17g %%.17g", (double)yr[q], (double)yi[q]);
    printf("\n");
    if constexpr (R >= 2) {
        constexpr int H = R / 2;
        real sr[H], si[H], dr[H], di[H];
        for (int q = 0; q < H; ++q) {
            sr[q] = xr[q] + xr[q + H]; si[q] = xi[q] + xi[q + H];
            dr[q] = xr[q] - xr[q + H]; di[q] = xi[q] - xi[q + H];
        }
        Dft<R>::run_pairs(sr, si, dr, di, yr, yi);
        printf("pairs %%d", R);
        for (int q = 0; q < R; ++q) printf(" %%.17g %%.17g", (double)yr[q], (double)yi[q]);
        printf("\n");
    }
}

int main(int argc, char** argv) {
    double in[64];
    for (int i = 0; i < 64; ++i) in[i] = atof(argv[1 + i]);
    one<2>(in); one<4>(in); one<8>(in); one<16>(in); one<32>(in);
    return 0;
}
"""


def _section():
    text = open(CORE).read()
    a, b = text.index("// [dft-begin]"), text.index("// [dft-end]")
    return text[a:b]


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("f64", [0, 1])
def test_small_dfts_on_the_host(tmp_path, fused, f64):
    src = tmp_path / "dft_host.cpp"
    exe = tmp_path / "dft_host"
    src.write_text(HARNESS % {"f64": f64, "fused": fused, "section": _section()})
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-o", str(exe), str(src)], check=True)
    rng = np.random.default_rng(20 + fused + 2 * f64)
    x = rng.standard_normal(64)
    out = subprocess.run([str(exe)] + [repr(float(v)) for v in x], check=True, capture_output=True, text=True).stdout
    z = x[0::2] + 1j * x[1::2]
    if not f64:
        z = z.astype(np.complex64).astype(np.complex128)
    eps = 1e-15 if f64 else 6e-8
    seen = 0
    for line in out.strip().splitlines():
        parts = line.split()
        kind, R = parts[0], int(parts[1])
        vals = np.array([float(v) for v in parts[2:]])
        y = vals[0::2] + 1j * vals[1::2]
        ref = np.fft.fft(z[:R])
        err = np.max(np.abs(y - ref)) / np.max(np.abs(ref))
        assert err < 8 * eps * max(1, np.log2(R)), (kind, R, err)
        seen += 1
    assert seen == 10


if __name__ == "__main__":
    sys.exit(pytest.main([__file__, "-q"]))
