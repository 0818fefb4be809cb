#!/usr/bin/env python3
"""numpy emulation of the cross-lane-paired (XL) plan: P = R = 16 (one butterfly per thread in every pass),
T = M/16 threads; the real-FFT partner of thread lane l lives in lane l ^ 32 of the same wave.  Development aid."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emulate_kernel import lds_phys, dft, pair_entry, pair_op
from pyaudiodsptools_amd import design


def jx_of(t, T):
    w, l = t >> 6, t & 63
    lo = 32 * w + (l & 31)
    j = np.where(l < 32, lo, T - lo)
    return np.where((w == 0) & (l == 32), T // 2, j)


def run_passes(reg, M, P, rads, inverse, jx):
    T = M // P
    tid = np.arange(T)
    S = 1
    lds = np.zeros(M, complex)
    NP = len(rads)
    for p, R in enumerate(rads):
        assert R == P
        paired = (p == 0) if inverse else (p == NP - 1)
        last = p == NP - 1
        j = jx if paired else tid
        u = reg.copy()
        if S > 1:
            jlo = j & (S - 1)
            u = u * np.exp(-2j * np.pi * np.outer(jlo, np.arange(R)) / (R * S))
        reg = dft(u, R)
        if last:
            break
        jlo = j & (S - 1)
        base = (j - jlo) * R + jlo
        for r in range(R):
            lds[lds_phys(base + r * S, R, S == 1)] = reg[:, r]
        next_paired = (not inverse) and (p + 1 == NP - 1)
        src = jx if next_paired else tid
        for q in range(P):
            reg[:, q] = lds[lds_phys(src + q * T, R, S == 1)]
        S *= R
    return reg


def emulate_block(window, H, M):
    P, R = 16, 16
    T = M // P
    D = M // R  # = T
    tid = np.arange(T)
    jx = jx_of(tid, T)
    assert sorted(jx.tolist()) == list(range(T))
    z = window[0::2] + 1j * window[1::2]
    reg = np.zeros((T, P), complex)
    for m in range(P):
        reg[:, m] = z[tid + T * m]
    reg = run_passes(reg, M, P, [16, 16, 16], False, jx)
    Z = np.fft.fft(z)
    for r in range(R):
        assert np.allclose(reg[:, r], Z[jx + D * r], atol=1e-6 * np.abs(Z).max())
    special = (tid == 0) | (tid == 32)
    # --- exchange registers 8..15 between lane halves: two v_permlane32_swap per register pair (i, i^1)
    part = tid ^ 32
    ex = reg.copy()
    for i in range(8, 16):
        ex[:, i] = np.where(special, reg[:, i], reg[part, i ^ 1])  # my reg i now holds partner's old reg i^1
    # --- 8 pair ops per regular thread: own reg r (k = jx + D r) with partner's old reg 15-r, found at my index (15-r)^1
    for t in range(T):
        if special[t]:
            continue
        for r in range(8):
            wc, g1, g2 = pair_entry(H, M, jx[t] + D * r)
            a, b = pair_op(ex[t, r], ex[t, (15 - r) ^ 1], wc, g1, g2)
            ex[t, r], ex[t, (15 - r) ^ 1] = a, b
    # special lanes keep all 16 of their own registers
    t = 0  # jx = 0: selfs r=0 (k=0), r=8 (k=M/2); pairs (r, 16-r)
    wc, g1, g2 = pair_entry(H, M, 0); ex[t, 0], _ = pair_op(ex[t, 0], ex[t, 0], wc, g1, g2)
    wc, g1, g2 = pair_entry(H, M, M // 2); ex[t, 8], _ = pair_op(ex[t, 8], ex[t, 8], wc, g1, g2)
    for r in range(1, 8):
        wc, g1, g2 = pair_entry(H, M, D * r)
        ex[t, r], ex[t, 16 - r] = pair_op(ex[t, r], ex[t, 16 - r], wc, g1, g2)
    t = 32  # jx = T/2: pairs (r, 15-r), k = T/2 + D r
    assert jx[t] == T // 2
    for r in range(8):
        wc, g1, g2 = pair_entry(H, M, T // 2 + D * r)
        ex[t, r], ex[t, 15 - r] = pair_op(ex[t, r], ex[t, 15 - r], wc, g1, g2)
    # --- swap back (same two swaps are an involution)
    reg2 = ex.copy()
    for i in range(8, 16):
        reg2[:, i] = np.where(special, ex[:, i], ex[part, i ^ 1])
    sw = reg2.imag + 1j * reg2.real
    sw = run_passes(sw, M, P, [16, 16, 16], True, jx)
    res = sw.imag + 1j * sw.real
    y = np.zeros(2 * M)
    for m in range(P):
        n = tid + T * m
        y[2 * n] = res[:, m].real
        y[2 * n + 1] = res[:, m].imag
    return y


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    M, N = 4096, 4096
    fir = design.FirStream(design.lowcut_kernel(800, 44100, N), N)
    geo = design.overlap_save_geometry(fir)
    Hf = design.engine_spectrum(fir, geo)
    H = (Hf[0::2] + 1j * Hf[1::2]).astype(complex)
    w = rng.uniform(-1, 1, 2 * M)
    y = emulate_block(w, H, M)
    ref = np.fft.irfft(np.fft.rfft(w) * H, 2 * M)
    print("XL M=4096 rel err", np.abs(y - ref).max() / np.abs(ref).max())
