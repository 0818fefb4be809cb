"""Host logic of the uniformly partitioned engines (csrc/adsp_upols.hip), checked on the CPU: the partitioning of the kernel
(design.partition_uniform) and a numpy mirror of upols_launch_pair's block bookkeeping - which blocks a call transforms, which it
multiplies and inverts, where their outputs land, how long the input history and the delay line must be - against the oracle's
float64 direct convolution.  The kernels themselves are checked on the GPU (tests/test_gpu_round5.py)."""
import numpy as np
import pytest

from oracle import fftfilter_oracle as orc
from pyaudiodsptools_amd import design


def floor_div(a, b):
    return a // b  # Python's // floors, like the C++ helper


class UpolsMirror:
    """upols_launch_pair in numpy: rfft / irfft stand in for the kernels, every index is the C++ code's."""

    def __init__(self, fir, block, max_steps):
        self.part = p = design.partition_uniform(fir, block)
        self.H = p.spectra[..., 0].astype(np.float64) + 1j * p.spectra[..., 1].astype(np.float64)
        self.N, self.B, self.P, self.delay, self.max_steps = int(fir.chunk_size), block, p.n_partitions, p.delay, max_steps
        self.nh = (2 * block + self.N - 1) // self.N
        self.R = (max_steps * self.N + self.delay + block - 1) // block + self.P + 3
        self.hist = np.zeros(self.nh * self.N)           # the ring: the last nh chunks
        self.zline = [(None, np.zeros(block + 1, complex)) for _ in range(self.R)]  # (block index held, spectrum)
        self.steps_done, self.fwd_done = 0, -1
        self.carry, self.carry_valid = np.zeros(0), False  # what the last block of the previous call holds beyond that call's end
        self.blocks_inverted = 0

    def _pair(self, x):
        N, B, n = self.N, self.B, len(x) // self.N
        t_call = self.steps_done * N
        t_end = t_call + n * N
        s = np.concatenate([self.hist, x])               # s[i] = sample t_call - nh N + i
        b_fwd_hi = floor_div(t_end, B) - 1
        for b in range(self.fwd_done + 1, b_fwd_hi + 1):
            rel = (b - 1) * B - t_call
            assert rel + self.nh * N >= 0, "window starts before the input history"
            lo = rel + self.nh * N
            win = s[lo:lo + 2 * B]
            assert len(win) == 2 * B, "window reaches beyond the input that has arrived"
            self.zline[b % self.R] = (b, np.fft.rfft(win))
        self.fwd_done = max(self.fwd_done, b_fwd_hi)
        b_lo, b_hi = floor_div(t_call - self.delay, B), floor_div(t_end - self.delay - 1, B)
        assert b_hi <= self.fwd_done
        assert self.fwd_done - (b_lo - self.P + 1) < self.R
        out = np.full(n * N, np.nan)
        use_carry = self.carry_valid and len(self.carry) > 0 and b_hi >= b_lo + 1
        if use_carry:                                    # the straddling block b_lo was computed whole by the previous call
            out[:len(self.carry)] = self.carry
        for b in range(b_lo + 1 if use_carry else b_lo, b_hi + 1):
            self.blocks_inverted += 1
            acc = np.zeros(B + 1, complex)
            for p in range(self.P):
                held, z = self.zline[(b - p) % self.R]
                assert held == b - p or (b - p < 0 and held is None), (b, p, held)  # no slot was overwritten too early
                acc += z * self.H[p]
            y = np.fft.irfft(acc, 2 * B)[B:]
            tau0 = b * B + self.delay - t_call            # rel_first + blk * B
            lo, hi = max(0, tau0), min(n * N, tau0 + B)
            out[lo:hi] = y[lo - tau0:hi - tau0]
            if b == b_hi:
                self.carry, self.carry_valid = y[n * N - tau0:].copy(), True   # (b_hi + 1) B + delay - t_end samples
                assert len(self.carry) == (b_hi + 1) * B + self.delay - t_end and 0 <= len(self.carry) < B
        assert not np.isnan(out).any(), "an output sample was produced by no block"
        cnt = min(n, self.nh)
        self.hist = np.concatenate([self.hist, x])[-self.nh * N:] if cnt else self.hist
        self.steps_done += n
        return out

    def apply(self, x):
        x = np.asarray(x, np.float64).reshape(-1)
        n = len(x) // self.N
        outs, done = [], 0
        while done < n:
            k = min(self.max_steps, n - done)
            outs.append(self._pair(x[done * self.N:(done + k) * self.N]))
            done += k
        return np.concatenate(outs)


@pytest.mark.parametrize("n,taps_len,block,max_steps,calls", [
    (88200, 44099, 8192, 1, [1, 1, 1]),            # the reference's Example4 low cut
    (88200, 88197, 8192, 2, [1, 2, 1]),            # its EQ composite: 11 partitions
    (88200, 88197, 16384, 1, [1, 1, 2]),           # the same on the larger block of the build: 6 partitions, history of one chunk
    (4096, 2500, 512, 3, [1, 3, 2, 5, 1]),         # small blocks: many blocks per call, calls of every length, sub-call splitting
    (1000, 1201, 256, 1, [1] * 7),                 # chunk not a multiple of the block, kernel longer than the chunk
    (520, 300, 256, 4, [2, 1, 4, 3]),              # history of several chunks (2 B > N)
])
def test_block_bookkeeping_matches_the_direct_convolution(n, taps_len, block, max_steps, calls):
    rng = np.random.default_rng(n + taps_len)
    taps = rng.standard_normal(taps_len) * np.hanning(taps_len) / taps_len ** 0.5
    lookahead = min(taps_len // 2, n - block - 3) if n - block - 3 > 0 else 0
    fir = design.FirStream(taps, n, latency_chunks=1 if n - lookahead >= block + 3 else 2, lookahead=lookahead)
    if fir.delay < block:
        fir = design.FirStream(taps, n, latency_chunks=-(-(block + lookahead) // n), lookahead=lookahead)
    eng = UpolsMirror(fir, block, max_steps)
    x = rng.uniform(-1, 1, sum(calls) * n)
    got, pos = [], 0
    for k in calls:
        got.append(eng.apply(x[pos * n:(pos + k) * n]))
        pos += k
    got = np.concatenate(got)
    ref = orc.direct_stream_convolution(taps, x, n, fir.latency_chunks, fir.lookahead)
    scale = np.abs(ref).max()
    assert scale > 0.05
    assert np.abs(got - ref).max() <= 2e-6 * scale   # complex64 spectra of the partitions: ~1e-7 relative


def test_partition_uniform_of_the_reference_shapes():
    for make, taps_expected, parts in ((lambda: design.lowcut_kernel(800, 44100, 88200), 44099, 6),
                                       (lambda: design.eq3_composite(100, 2, 700, -4, 8000, 5, 44100, 88200), 88197, 11)):
        fir = design.FirStream(make(), 88200)
        assert len(fir.taps) == taps_expected
        p = design.partition_uniform(fir, 8192)
        assert p.n_partitions == parts and p.delay % 4 == 0 and p.delay + p.shift == fir.delay and p.delay >= 8192
        assert p.spectra.shape == (parts, 8193, 2) and p.spectra.dtype == np.float32
        # the pieces put back together are the (delayed) kernel
        spec = p.spectra[..., 0].astype(np.float64) + 1j * p.spectra[..., 1]
        pieces = np.fft.irfft(spec, 16384, axis=1)[:, :8192].reshape(-1)
        full = np.concatenate([np.zeros(p.shift), fir.taps])
        assert np.abs(pieces[:len(full)] - full).max() <= 1e-6 * np.abs(full).max() and np.abs(pieces[len(full):]).max() <= 1e-6 * np.abs(full).max()
    with pytest.raises(ValueError):
        design.partition_uniform(design.FirStream(np.ones(40000), 30002), 8192)   # chunk not a multiple of 4
    with pytest.raises(ValueError):
        design.partition_uniform(design.FirStream(np.ones(40000), 4096), 8192)    # delayed by less than a block


def test_block_size_policy():
    """design.choose_uniform_block: the larger block only for kernels of many partitions, where the delay allows it and a call has blocks
    enough to fill the chip."""
    lc = design.FirStream(design.lowcut_kernel(800, 44100, 88200), 88200)                      # 6 partitions of 8192
    eq = design.FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, 44100, 88200), 88200)  # 11
    assert all(design.choose_uniform_block(lc, c) == 8192 for c in (1, 16)) and all(design.choose_uniform_block(lc, c) == 16384 for c in (32, 64, 256, 1024, 4096))
    assert design.choose_uniform_block(eq, 1) == 8192 and design.choose_uniform_block(eq, 16) == 8192
    assert design.choose_uniform_block(eq, 32) == 16384 and design.choose_uniform_block(eq, 64) == 16384 and design.choose_uniform_block(eq, 4096) == 16384
    three = design.FirStream(np.ones(20000), 88200)                                             # 3 partitions of 8192: the small block everywhere
    assert all(design.choose_uniform_block(three, c) == 8192 for c in (64, 4096))
    short_delay = design.FirStream(np.ones(100000), 20000, latency_chunks=1, lookahead=8000)   # delayed by 12000 samples
    assert design.choose_uniform_block(short_delay, 4096) == 8192
    with pytest.raises(ValueError):
        design.partition_uniform(short_delay, 16384)
    assert design.partition_uniform(lc, 16384).n_partitions == 3 and design.partition_uniform(lc, 8192).n_partitions == 6
