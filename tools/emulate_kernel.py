#!/usr/bin/env python3
"""numpy emulation of fftconv_kernel.hpp's per-thread algorithm (development aid, not product code).

Vectorised over threads: registers are arrays [T, P].  Mirrors Plan/Pass/spectrum_stage/pair_op and
the host table builders of adsp_capi.hip so that indexing and algebra can be checked without a GPU.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyaudiodsptools_amd import design

PLANS = {64: (16, [8, 8]), 128: (16, [16, 8]), 256: (16, [4, 8, 8]), 512: (16, [16, 4, 8]), 1024: (16, [16, 8, 8]),
         2048: (16, [16, 16, 8]), 4096: (32, [16, 16, 16]), 8192: (32, [32, 16, 16]), 16384: (32, [32, 2, 16, 16])}


def lds_phys(a, R, swz):
    if not swz:
        return a
    mask = min(R, 16) - 1
    sh = 4 if R <= 16 else 5
    return a ^ ((a >> sh) & mask)


def dft(x, R):  # x [..., R] complex, forward
    k = np.arange(R)
    W = np.exp(-2j * np.pi * np.outer(k, k) / R)
    return x @ W.T


def run_passes(reg, M, P, rads, inverse, ja, jb):
    """reg [T,P] complex (already 'swapped view' if inverse). Forward-sign Stockham passes."""
    T = M // P
    tid = np.arange(T)
    NP = len(rads)
    S = 1
    lds = np.zeros(M, complex)
    for p, R in enumerate(rads):
        NB = P // R
        paired = (p == 0) if inverse else (p == NP - 1)
        last = p == NP - 1
        js = []
        for i in range(NB):
            j = (ja if i == 0 else jb) if paired else tid + T * i
            js.append(j)
            u = reg[:, i + np.arange(R) * NB].copy()
            if S > 1:
                jlo = j & (S - 1)
                q = np.arange(R)
                u = u * np.exp(-2j * np.pi * np.outer(jlo, q) / (R * S))
            reg[:, i + np.arange(R) * NB] = dft(u, R)
        if last:
            break
        for i in range(NB):
            j = js[i]
            jlo = j & (S - 1)
            base = (j - jlo) * R + jlo
            for r in range(R):
                a = lds_phys(base + r * S, R, S == 1)
                lds[a] = reg[:, i + r * NB]
        assert len(np.unique(np.concatenate([lds_phys((js[i] - (js[i] & (S - 1))) * R + (js[i] & (S - 1)) + r * S, R, S == 1) for i in range(NB) for r in range(R)]))) == M
        next_paired = (not inverse) and (p + 1 == NP - 1)
        if not next_paired:
            for m in range(P):
                reg[:, m] = lds[lds_phys(tid + T * m, R, S == 1)]
        else:
            Rn = P // 2
            for q in range(Rn):
                reg[:, 2 * q] = lds[lds_phys(ja + q * (M // Rn), R, S == 1)]
                reg[:, 2 * q + 1] = lds[lds_phys(jb + q * (M // Rn), R, S == 1)]
        S *= R
    return reg


def pair_entry(H, M, k):
    """(c1, c2, c4) of the 2x2 pairing matrix (adsp_capi.hip::pair_entry)."""
    ang = np.pi * k / M
    sc = 1.0 / (4.0 * M)
    wc = complex(-np.sin(ang), -np.cos(ang))
    g1, g2 = H[k] * sc, np.conj(H[M - k]) * sc
    s, d = g1 + g2, g1 - g2
    return 2 * s + 2 * d * wc.real, -2j * d * wc.imag, 2 * s - 2 * d * wc.real


def pair_op(za, zb, c1, c2, c4):
    """Zy[k] = c1 Za + c2 conj(Zb);  Zy[M-k] = conj(c4 conj(Zb) - c2 Za)   (fftconv_kernel.hpp::pair_op)."""
    return c1 * za + c2 * np.conj(zb), np.conj(c4 * np.conj(zb) - c2 * za)


def emulate_block(window, H, M):
    """window: 2M real samples; H: M+1 complex bins. Returns the 2M circular-convolution samples."""
    P, rads = PLANS[M]
    T = M // P
    R = P // 2
    tid = np.arange(T)
    z = window[0::2] + 1j * window[1::2]
    reg = np.zeros((T, P), complex)
    for m in range(P):
        reg[:, m] = z[tid + T * m]
    ja = tid.copy()
    jb = np.where(tid == 0, T, 2 * T - tid)
    reg = run_passes(reg, M, P, rads, False, ja, jb)
    # check: reg[:,2r] = Z[ja + 2T r], reg[:,2r+1] = Z[jb + 2T r]
    Z = np.fft.fft(z)
    for r in range(R):
        assert np.allclose(reg[:, 2 * r], Z[ja + 2 * T * r], atol=1e-6 * np.abs(Z).max()), "fwd a"
        assert np.allclose(reg[:, 2 * r + 1], Z[jb + 2 * T * r], atol=1e-6 * np.abs(Z).max()), "fwd b"
    # spectrum stage
    for t in range(T):
        if t != 0:
            for r in range(R):
                wc, g1, g2 = pair_entry(H, M, t + 2 * T * r)
                a, b = pair_op(reg[t, 2 * r], reg[t, 2 * (R - 1 - r) + 1], wc, g1, g2)
                reg[t, 2 * r], reg[t, 2 * (R - 1 - r) + 1] = a, b
        else:
            wc, g1, g2 = pair_entry(H, M, 0)
            reg[0, 0], _ = pair_op(reg[0, 0], reg[0, 0], wc, g1, g2)
            wc, g1, g2 = pair_entry(H, M, M // 2)
            reg[0, R], _ = pair_op(reg[0, R], reg[0, R], wc, g1, g2)
            for r in range(1, R // 2):
                wc, g1, g2 = pair_entry(H, M, 2 * T * r)
                a, b = pair_op(reg[0, 2 * r], reg[0, 2 * (R - r)], wc, g1, g2)
                reg[0, 2 * r], reg[0, 2 * (R - r)] = a, b
            for r in range(R // 2):
                wc, g1, g2 = pair_entry(H, M, T + 2 * T * r)
                a, b = pair_op(reg[0, 2 * r + 1], reg[0, 2 * (R - 1 - r) + 1], wc, g1, g2)
                reg[0, 2 * r + 1], reg[0, 2 * (R - 1 - r) + 1] = a, b
    # inverse = forward on swapped parts
    sw = reg.imag + 1j * reg.real
    sw = run_passes(sw, M, P, rads[::-1], True, ja, jb)
    res = sw.imag + 1j * sw.real
    y = np.zeros(2 * M)
    for m in range(P):
        n = tid + T * m
        y[2 * n] = res[:, m].real
        y[2 * n + 1] = res[:, m].imag
    return y


def main():
    rng = np.random.default_rng(0)
    for M in sorted(PLANS):
        F = 2 * M
        for N in (F // 2, F // 4):
            if N < 64 or N > 8192:
                continue
            fir = design.FirStream(design.lowcut_kernel(800, 44100, N), N)
            if F == 4 * N:
                eq = design.FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, 44100, N), N)
                fir = fir.then(eq).then(design.FirStream(design.highcut_kernel(8000, 44100, N), N))
            geo = design.overlap_save_geometry(fir)
            assert geo.fft_size == F, (geo, F)
            Hf = design.engine_spectrum(fir, geo)
            H = (Hf[0::2] + 1j * Hf[1::2]).astype(complex)
            w = rng.uniform(-1, 1, F)
            y = emulate_block(w, H, M)
            ref = np.fft.irfft(np.fft.rfft(w) * H, F)
            err = np.abs(y - ref).max() / np.abs(ref).max()
            print(f"M={M:6d} N={N:5d} rel err {err:.2e}")
            assert err < 1e-9


if __name__ == "__main__":
    main()
