#!/usr/bin/env python3
"""(CPU) checks for the 3 * 2^k plan of fftconv_kernel.hpp: the prime-factor DFT-12 index maps, and the LDS bank behaviour of
every exchange of Plan<3072, 48, 16 x 16 x 12> (half-buffer rounds included) under the swizzle of lds_phys.

Bank model (MI355X_MICROARCH.md, LDS): ds_write_b64 is served in 4 groups of 16 contiguous lanes, bank = (byte address / 4)
mod 32, i.e. 8-byte element index mod 16; ds_read_b64 in 2 groups of 32 lanes, element index mod 32.  Prints the worst
number of distinct addresses on one bank per lane group (1 = conflict-free)."""
import numpy as np


def dft12_pfa(x):
    a = np.zeros((3, 4), complex)
    w3 = np.exp(-2j * np.pi / 3)
    for n2 in range(4):
        v = [x[(3 * n2) % 12], x[(4 + 3 * n2) % 12], x[(8 + 3 * n2) % 12]]
        for k1 in range(3):
            a[k1, n2] = sum(v[n1] * w3 ** (n1 * k1) for n1 in range(3))
    y = np.zeros(12, complex)
    for k1 in range(3):
        b = np.fft.fft(a[k1])
        for k2 in range(4):
            y[(4 * k1 + 9 * k2) % 12] = b[k2]
    return y


x = np.random.default_rng(0).standard_normal(12) + 1j * np.random.default_rng(1).standard_normal(12)
assert np.allclose(dft12_pfa(x), np.fft.fft(x)), "DFT-12 prime-factor maps"
print("DFT-12 (3 x 4 prime factor): index maps OK")


def lds_phys(a, R, swz):
    if not swz:
        return a
    mask = 3 if R == 12 else min(R, 16) - 1
    sh = 4 if R <= 16 else 5
    return a ^ ((a >> sh) & mask)


def worst(addr, group, banks):
    """addr [lanes]: element index per lane; lanes served in groups; distinct addresses per bank."""
    w = 0
    for g0 in range(0, len(addr), group):
        a = np.unique(addr[g0:g0 + group])
        _, cnt = np.unique(a % banks, return_counts=True)
        w = max(w, cnt.max())
    return w


def paired_bfly(i, tid, T, NBL):
    u = i >> 1
    if i & 1 == 0:
        return u * T + tid
    return np.where((u == 0) & (tid == 0), NBL * T // 2, (NBL - u) * T - tid)


def check(M, P, fwd, half):
    T = M // P
    tid = np.arange(T)
    NP = len(fwd)
    RL = fwd[-1]
    NBL = P // RL
    for inverse in (False, True):
        rads = fwd[::-1] if inverse else fwd
        S = 1
        for p, R in enumerate(rads[:-1]):
            NB = P // R
            paired = inverse and p == 0
            swz = S == 1
            seen = []
            ww = 0
            for i in range(NB):
                j = paired_bfly(i, tid, T, NBL) if paired else tid + T * i
                jlo = j % S
                base = (j - jlo) * R + jlo
                for r in range(R):
                    a = base + r * S
                    for h in ((0, 1) if half else (None,)):
                        sel = np.ones(T, bool) if h is None else ((base >= M // 2) == (h == 1))
                        if not sel.any():
                            continue
                        ph = lds_phys(a - (0 if not h else M // 2), R, swz)
                        # lanes that do not take part in this round keep their slot in the 16-lane group idle
                        for g0 in range(0, T, 16):
                            s = sel[g0:g0 + 16]
                            if s.any():
                                ww = max(ww, worst(ph[g0:g0 + 16][s], 16, 16))
                    seen.append(lds_phys(a, R, swz) if not half else a)
            allw = np.concatenate(seen)
            assert len(np.unique(allw)) == M, "every element written once"
            next_paired = (not inverse) and p + 1 == NP - 1
            rw = 0
            if not next_paired:
                for m in range(P):
                    e = tid + T * m
                    rw = max(rw, worst(lds_phys(e % (M // 2) if half else e, R, swz), 32, 32))
            else:
                for q in range(RL):
                    for i in range(NBL):
                        e = paired_bfly(i, tid, T, NBL) + q * (M // RL)
                        rw = max(rw, worst(lds_phys(e % (M // 2) if half else e, R, swz), 32, 32))
            print(f"M={M} {'inv' if inverse else 'fwd'} pass {p}: R={R:2d} S={S:4d} swizzle={int(swz)}  write worst {ww}-way  read worst {rw}-way")
            S *= R


check(3072, 48, [16, 16, 12], True)
check(3072, 24, [8, 8, 4, 12], False)
check(3072, 24, [8, 8, 4, 12], True)
check(4096, 16, [16, 16, 16], False)
