"""Host-side filter design (float64) and overlap-save geometry.

Everything here runs once per device construction, like the reference's ``__init__`` bodies
(pyAudioDspTools/EffectFFTFilter.py:18-47, :91-123; EffectEQ3BandFFT.py:47-153).  The reference
keeps four/one 3N-point complex spectra and recombines them on every call; here each device is
reduced to ONE time-domain FIR whose 2N-point real spectrum is handed to the
GPU engine.
"""
from dataclasses import dataclass

import numpy as np


def filter_length(chunk_size):
    """L = N//2 - 1 (EffectFFTFilter.py:22); d = L//2 is the look-ahead of the kept slice (the slice starts at N + L//2,
    EffectFFTFilter.py:24: (L-1)/2 for the odd L every chunk size with N % 4 == 0 gives, L/2 for the even L of N = 30, 1002 ..)."""
    taps = int(chunk_size) // 2 - 1
    if taps < 1:
        raise ValueError(f"chunk_size {chunk_size}: the reference's filter length N//2 - 1 must be at least 1")
    return taps, taps // 2


def _lowpass(cutoff_hz, fs, taps, window):
    n = np.arange(taps, dtype=np.float64) - (taps - 1) / 2
    h = np.sinc(2 * cutoff_hz / fs * n) * window
    return h / h.sum()


def _convolve(a, b):
    """Linear convolution of two kernels in float64: numpy's direct sum for short ones, an FFT product (zero-padded to a power of two; error
    ~1e-15 of the peak, the design goldens hold 1e-12) where the direct sum would take seconds - the EQ's mid band at chunk 88200 is
    44099 x 44099 taps, 3.4 s of a 3.5 s constructor."""
    if min(len(a), len(b)) <= 8192:
        return np.convolve(a, b)
    n = len(a) + len(b) - 1
    f = 1 << (n - 1).bit_length()
    return np.fft.irfft(np.fft.rfft(a, f) * np.fft.rfft(b, f), f)[:n]


def _invert(h):
    g = -h
    g[(len(h) - 1) // 2] += 1.0
    return g


def highcut_kernel(cutoff_hz, fs, chunk_size):
    """Blackman windowed-sinc low-pass, unity DC gain (EffectFFTFilter.py:28-37)."""
    taps, _ = filter_length(chunk_size)
    return _lowpass(cutoff_hz, fs, taps, np.blackman(taps))


def lowcut_kernel(cutoff_hz, fs, chunk_size):
    """Spectral inversion of the low-pass (EffectFFTFilter.py:101-113)."""
    return _invert(highcut_kernel(cutoff_hz, fs, chunk_size))


def eq3_kernels(lowshelf_hz, midband_hz, highshelf_hz, fs, chunk_size):
    """The four Kaiser(6.0) kernels of EffectEQ3BandFFT.py:70-133 with their 0.75x / 1.25x cutoffs."""
    taps, _ = filter_length(chunk_size)
    w = np.kaiser(taps, 6.0)
    return dict(
        highshelf=_invert(_lowpass(highshelf_hz - highshelf_hz / 4, fs, taps, w)),
        lowshelf=_lowpass(lowshelf_hz + lowshelf_hz / 4, fs, taps, w),
        mid_lowpass=_lowpass(midband_hz + midband_hz / 4, fs, taps, w),
        mid_highpass=_invert(_lowpass(midband_hz - midband_hz / 4, fs, taps, w)),
    )


def eq3_composite(lowshelf_hz, lowshelf_db, midband_hz, midband_db, highshelf_hz, highshelf_db, fs, chunk_size):
    """One (2L-1)-tap FIR equal to EffectEQ3BandFFT.apply (:179-209).

    sum_b (g_b - 1) band_b + dry, with the mid band = highpass (*) lowpass (spectrum product at :188,
    hence centred at 2d - the reference's extra mid-band delay is preserved) and dry = delta at d.
    """
    taps, d = filter_length(chunk_size)
    k = eq3_kernels(lowshelf_hz, midband_hz, highshelf_hz, fs, chunk_size)
    c = np.zeros(2 * taps - 1)
    c[:taps] += (10 ** (highshelf_db / 20) - 1) * k["highshelf"]
    c[:taps] += (10 ** (lowshelf_db / 20) - 1) * k["lowshelf"]
    c += (10 ** (midband_db / 20) - 1) * _convolve(k["mid_highpass"], k["mid_lowpass"])
    c[d] += 1.0
    return c


def reference_spectrum_3n(kernel, chunk_size):
    """The reference's inspectable ``sinc_filter`` attribute: fft of the kernel zero-padded to 3N."""
    buf = np.zeros(3 * int(chunk_size))
    buf[: len(kernel)] = kernel
    return np.fft.fft(buf)


TRIM_EPS = 1e-8  # FirStream.trimmed: relative weight (of sum|taps|) a fused chain may drop at its two ends


@dataclass
class FirStream:
    """out[tau] = sum_t taps[t] * s[tau - latency_chunks*N + lookahead - t]  (zero history)."""
    taps: np.ndarray
    chunk_size: int
    latency_chunks: int = 1
    lookahead: int = None

    def __post_init__(self):
        self.taps = np.asarray(self.taps, dtype=np.float64)
        if self.lookahead is None:
            self.lookahead = filter_length(self.chunk_size)[1]

    @property
    def delay(self):
        return self.latency_chunks * self.chunk_size - self.lookahead

    def then(self, other):
        """Series connection: kernels convolve, delays add (chain LowCut -> EQ -> HighCut)."""
        assert other.chunk_size == self.chunk_size
        return FirStream(_convolve(self.taps, other.taps), self.chunk_size,
                         self.latency_chunks + other.latency_chunks, self.lookahead + other.lookahead)

    def trimmed(self, eps=TRIM_EPS):
        """Drop leading / trailing taps whose absolute sum stays below eps/2 * sum|taps| per side.

        The ends of a product of windowed sincs are many orders of magnitude below float32 resolution: for the
        LowCut -> EQ3 -> HighCut chain at N = 8192 (16377 taps) 6976 end taps together weigh 1e-8 of sum|taps|, so leaving
        them out changes any output by at most 1e-8 * sum|taps| * max|x| (~6e-8 of full scale - a tenth of the float32
        FFT pipeline's own rounding, 1/300 of the 1e-5 parity budget), while the shorter kernel lets a 4N transform keep
        2.75 N instead of 2 N samples.  The dropped leading taps move into the delay (lookahead shrinks)."""
        a = np.abs(self.taps)
        budget = 0.5 * float(eps) * a.sum()
        lo = int(np.searchsorted(np.cumsum(a), budget, side="right"))            # taps[:lo] weigh <= budget
        hi = len(a) - int(np.searchsorted(np.cumsum(a[::-1]), budget, side="right"))
        if (lo == 0 and hi == len(a)) or hi - lo < 1:  # nothing to drop / a budget that would drop everything
            return self
        return FirStream(self.taps[lo:hi], self.chunk_size, self.latency_chunks, self.lookahead - lo)


@dataclass
class Geometry:
    fft_size: int
    history_chunks: int
    lookback: int
    out_offset: int
    shift: int           # circular index of taps[0] in the F-point kernel buffer: > 0 delays the kernel so that
                         # out_offset is aligned, < 0 (zero_phase) centres a symmetric kernel on index 0
    max_block_outputs: int
    zero_phase: bool = False  # the spectrum is real: the engine takes its 3-constants-per-bin-pair path


def overlap_save_geometry(fir: FirStream, fft_mult: float = 0, optimize_for: str = "stream") -> Geometry:
    """Choose F, the window position and the kept slice for the GPU engine (include/adsp.h).

    Output tau is y[tau - D] with y = taps (*) s and D = delay.  The window for the block starting at
    output-time o begins at input-time o - lookback, so output tau sits at circular index
    (tau - o) + lookback - D + shift; it is wrap-free when lookback >= D + len(taps) - 1.
    Everything is kept a multiple of N/4 (>= 2 * threads-per-transform for every plan).
    fft_mult = 4 forces a 4N transform (fewer, larger blocks in multi-step launches), 1.5 / 2 likewise; optimize_for="batch"
    picks 4N by itself when a 2N transform keeps only half of its samples (EQ: measured +15 % in multi-step launches).
    """
    n = int(fir.chunk_size)
    if n < 64 or n > 8192 or n & (n - 1):
        return _generic_geometry(fir, fft_mult, optimize_for)
    g = n // 4
    m = len(fir.taps)
    d_total = fir.delay
    if d_total <= 0:
        raise ValueError("non-causal stream")
    lookback = -(-(d_total + m - 1) // g) * g
    centre = (m - 1) // 2
    symmetric = m % 2 == 1 and np.abs(fir.taps - fir.taps[::-1]).max() <= 1e-13 * np.abs(fir.taps).max()
    zero_phase = bool(symmetric and (d_total + centre) % g == 0)
    if zero_phase:
        # A symmetric kernel centred on circular index 0 has a REAL spectrum (half the spectrum-stage arithmetic and
        # table traffic in the kernel).  Valid circular indices are centre .. F-1-centre; the kept slice starts at
        # lookback - delay - centre, a multiple of N/4 whenever delay + centre is (the reference's low/high cut
        # filters: delay = N - d, centre = d).
        shift = -centre
    else:
        shift = (-(lookback - d_total)) % g
    out_offset = lookback - d_total + shift
    if not fft_mult and optimize_for == "batch" and out_offset + n <= 4 * n - max(0, -shift):
        # Multi-step launches: a 4N transform keeps a larger share of its samples (cut filters: 3.5 N of 4 N against 1.5 N of
        # 2 N; EQ: 3 N against N).  It is chosen when the 2N transform would keep a chunk or less (measured +15 %), and for
        # N = 1024 .. 4096 whenever it fits: there the 4N transform runs on the M = 2048 .. 8192 plans, which are as fast per
        # point as the 2N ones (measured on MI355X, profiles/r3_shapes_4n.txt: +13 % at N = 4096, +7 % at 1024, +4 % at 2048;
        # -4 % at N = 512, and N = 8192 would need the slow M = 16384 plan).
        if (2 * n - max(0, -shift) - out_offset) // g * g <= n or n in (1024, 2048, 4096):
            fft_mult = 4
    # fft_mult = 1.5 (F = 3 * 2^k, where libadsp has such a plan: N = 4096) is all a single-step launch of the reference's cut
    # filters needs (N + 2d samples): 27 % less butterfly work per kept chunk than a 2N transform.  It is opt-in: its plan runs
    # one wave per transform and only beats the 2N plan with >= 8192 channels per GPU.
    candidates = (2 * n, 4 * n) if not fft_mult else (int(round(fft_mult * n)),)
    for f in candidates:
        if out_offset + n <= f - max(0, -shift):
            break
    else:
        raise ValueError(f"kernel of {m} taps does not fit a {fft_mult or 4}N transform at N={n}")
    hist = -(-lookback // n)
    vmax = ((f - max(0, -shift) - out_offset) // g) * g
    return Geometry(f, hist, lookback, out_offset, shift, vmax, zero_phase)


PCM16_GAIN = 32767.0 / 32768.0  # int16 -> float (/32768, Utility.py:237) and float -> int16 (*32767, Utility.py:306)


def _generic_geometry(fir: FirStream, fft_mult: int, optimize_for: str) -> Geometry:
    """Any chunk size divisible by 4: the transform is not tied to the chunk.

    Blocks of V kept samples tile the time axis; V and out_offset are whole register-pair segments (4T samples), the
    window start a multiple of 4 samples (the kernel moves 16 bytes per access), which a 0..3-tap delay of the
    kernel (`shift`) arranges.  stream: the smallest transform that returns a whole chunk per block;
    batch: the transform with the fewest flops per kept sample."""
    from . import _capi
    n, m, d_total = int(fir.chunk_size), len(fir.taps), fir.delay
    if n < 4:
        raise ValueError(f"chunk_size {n}: need at least 4 samples")
    # chunk sizes that are not multiples of 4 (or shorter than 16) run on the kernel's dword-access form (round 4): nothing on
    # the time axis has to be aligned there
    align = 1 if (n % 4 or n < 16) else 4
    if d_total <= 0:
        raise ValueError("non-causal stream")
    centre = (m - 1) // 2
    symmetric = m % 2 == 1 and np.abs(fir.taps - fir.taps[::-1]).max() <= 1e-13 * np.abs(fir.taps).max()
    # a symmetric kernel centred on circular index 0 (real spectrum, see overlap_save_geometry) also needs less room
    # in front of the kept slice: (m-1)/2 wrapped taps instead of m-1
    zero_phase = bool(symmetric and (d_total + centre) % align == 0)
    best = None
    for log_f in range(7, 16):
        f = 1 << log_f
        if fft_mult and f != fft_mult * n:
            continue
        try:
            # register PAIRS are stored together (16-byte accesses): kept ranges are multiples of 4T samples
            t2 = 4 * _capi.plan_describe(n, f)["threads_per_transform"]
        except _capi.AdspError:
            continue
        if zero_phase:
            j0 = max(t2, -(-centre // t2) * t2)
            v = (f - centre - j0) // t2 * t2
            shift = -centre
        else:
            j0 = -(-(m + 2) // t2) * t2
            v = (f - j0) // t2 * t2
            shift = (d_total + j0) % align
        if v < t2:
            continue
        lookback = d_total + j0 - shift
        hist = -(-lookback // n)
        if hist > _capi.ADSP_MAX_HISTORY:
            continue
        cost = f * log_f / v
        geo = Geometry(f, hist, lookback, j0, shift, v, zero_phase)
        if optimize_for == "stream":
            if v >= n:          # one block per call: smallest such transform wins
                return geo
            if best is None or cost < best[0]:
                best = (cost, geo)
        elif best is None or cost < best[0]:
            best = (cost, geo)
    if best is None:
        raise ValueError(f"a kernel of {m} taps at chunk_size {n} does not fit a 32768-point transform "
                         "(partitioned convolution is not implemented)")
    return best[1]


def fits_one_transform(fir: FirStream) -> bool:
    try:
        overlap_save_geometry(fir)
        return True
    except ValueError:
        return False


def partition(fir: FirStream, max_taps: int = 14336):
    """Split a long kernel into parts that each fit one transform: out = sum_p conv(part_p, x delayed by p*max_taps).

    Part p keeps the chunk size and latency and moves the look-ahead back by p*max_taps samples (delays add), so every
    part is an ordinary FirStream for its own engine; results are summed (adsp_set_accumulate)."""
    m = len(fir.taps)
    parts = []
    for p in range(-(-m // max_taps)):
        sl = fir.taps[p * max_taps:(p + 1) * max_taps]
        parts.append(FirStream(sl, fir.chunk_size, fir.latency_chunks, fir.lookahead - p * max_taps))
    return parts


@dataclass
class UniformPartition:
    """A long FIR cut for the uniformly partitioned engines (include/adsp.h, adsp_upols_*): out[tau] = y[tau - delay] with
    y = (shift zeros + taps) (*) s, the delayed kernel cut into n_partitions pieces of `block` taps."""
    block: int
    n_partitions: int
    delay: int          # multiple of 4
    shift: int          # 0..3 taps of extra kernel delay that make `delay` a multiple of 4
    spectra: np.ndarray  # float32 [n_partitions, block + 1, 2]: rfft of each piece zero-padded to 2 * block


def choose_uniform_block(fir: "FirStream", channels: int, sizes=(8192, 16384)) -> int:
    """Block size of a uniformly partitioned engine.  Per output sample the multiply launch reads taps / B x 16 bytes (8 of tables,
    8 of spectra), so a larger block pays for kernels of many partitions - where the stream's delay allows it (partition_uniform:
    delay >= block) and a call has workgroups enough to occupy the chip; below that the call is as long as ONE workgroup's chain of
    work, and the shorter workgroups of the small block win.  Measured at chunk 88200 (low cut: 6 partitions of 8192, EQ composite: 11)
    since blocks of 16384 run on 512 threads (tools/sessions/r6_session36.sh, 37: profiles/r6f_upols_block_sizes.txt), us per call,
    blocks of 8192 -> 16384:  16 channels 29.5 -> 33 / 35.6 -> 39.3;  32 channels 39.5 -> 35.8 / 50.3 -> 42.1;  64 channels 53.4 -> 52.0 /
    70.5 -> 65.6;  256 channels 226 -> 193 / 319 -> 257;  1024 channels 806 -> 703 / 1150 -> 946.  (On 64 points per thread in 256
    threads - rounds 5, 6a - the large block won from 256 channels on only: profiles/r6_upols_block_sizes.txt.)"""
    delay = int(fir.delay) - int(fir.delay) % 4
    valid = sorted(b for b in sizes if b <= delay)
    if not valid:
        return min(sizes)  # (partition_uniform raises for it)
    small, big = valid[0], valid[-1]
    many_partitions = -(-(len(fir.taps) + int(fir.delay) % 4) // small) >= 4
    return big if many_partitions and int(channels) * int(fir.chunk_size) >= 128 * big else small


def partition_uniform(fir: FirStream, block: int, gain: float = 1.0) -> UniformPartition:
    """Cut `fir` into partitions of `block` taps.  The stream delay fir.delay (= latency_chunks * N - lookahead) is rounded DOWN to a
    multiple of 4 samples - the kernels move four samples per access - by delaying the kernel by the remainder instead (0..3 leading
    zero taps).  Raises ValueError where the engines' conditions do not hold (chunk not a multiple of 4, delay shorter than a block:
    an output block could then need input that has not arrived)."""
    n = int(fir.chunk_size)
    if n % 4 or n < 16:
        raise ValueError(f"chunk_size {n}: the partitioned engines need a multiple of 4, >= 16")
    d_total = int(fir.delay)
    shift = d_total % 4
    delay = d_total - shift
    if delay < block:
        raise ValueError(f"stream delay {d_total} is shorter than a block of {block} samples")
    taps = np.concatenate([np.zeros(shift), np.asarray(fir.taps, dtype=np.float64) * gain])
    parts = -(-len(taps) // block)
    padded = np.zeros(parts * block)
    padded[:len(taps)] = taps
    spec = np.fft.rfft(padded.reshape(parts, block), 2 * block, axis=1).astype(np.complex64)
    return UniformPartition(int(block), int(parts), int(delay), int(shift), np.ascontiguousarray(spec).view(np.float32).reshape(parts, block + 1, 2))


def engine_spectrum(fir: FirStream, geo: Geometry, gain: float = 1.0, dtype=np.float32) -> np.ndarray:
    """rfft of the (shift-delayed) kernel at F points, float64 -> complex64, as interleaved float32
    (dtype=np.float64: complex128 as interleaved float64, for the float64 engines).

    `gain` scales the kernel; int16 engines fold the reference's two PCM conversions into it (PCM16_GAIN)."""
    padded = np.zeros(geo.fft_size)
    padded[(geo.shift + np.arange(len(fir.taps))) % geo.fft_size] = fir.taps * gain
    spec = np.fft.rfft(padded)
    if geo.zero_phase:
        spec = spec.real + 0j  # circularly even kernel: the imaginary part is round-off, and exact zeros select the real path
    if dtype == np.float64:
        return np.ascontiguousarray(spec.astype(np.complex128)).view(np.float64)
    return np.ascontiguousarray(spec.astype(np.complex64)).view(np.float32)
