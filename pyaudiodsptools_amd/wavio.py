"""Chunk / WAV plumbing either side of the FFT filters, rebuilt around the GPU engine.

Host helpers with the reference's names and semantics
(pyAudioDspTools/Utility.py:8-48 MakeChunks/CombineChunks, :195-312 WAV import/export), plus `WavBank`:
many 16-bit WAV files -> ONE int16 device batch [steps, channels, chunk] -> filtered by an int16
engine (conversion fused into the kernel, half the HBM traffic of the float path) -> int16 -> WAV files.
This is Example1.py / Example2.py for many files at once.

Output samples of the float32 FFT engine can differ from the reference's by one LSB: `(y * 32767).astype(int16)`
truncates, and the float32 filter output differs from numpy's in the last bits - measured: 0.02 % of the samples of the
reference's own WAV, 0.22 % of full-scale noise (profiles/r2_pcm16_histogram.json).  `WavBank.process(fir, exact=True)`
is bit-identical (float64 direct sum, adsp_exact_*).
"""
import math
import wave

import numpy as np

from . import config
from .design import FirStream
from .effects import MixSignals, VolumeChange  # noqa: F401  (the rest of the reference's Utility.py: this module is `Utility`)
from .engine import ExactFirEngine, FirEngine, make_engine
from .signals import (Convert16BitTodBV, ConvertdBVTo16Bit, Dither16BitTo8Bit, Dither32BitIntTo16BitInt,  # noqa: F401
                      InfodBV, InfodBV16Bit)


# ---- the reference's chunk plumbing ---------------------------------------------------------
def MakeChunks(float32_array_input):
    """Pad to a multiple of config.chunk_size and split (Utility.py:22-28, including its pad test on
    `len % number_of_chunks`)."""
    n = config.chunk_size
    number_of_chunks = math.ceil(np.float32(len(float32_array_input) / n))
    if len(float32_array_input) % number_of_chunks != 0:
        pad = n - (len(float32_array_input) % n)
        float32_array_input = np.append(float32_array_input, np.zeros(pad, dtype="float32"))
    return np.split(float32_array_input, number_of_chunks)


def CombineChunks(float_array_input):
    """Re-join chunks into one float32 array (Utility.py:45-48, without the quadratic append)."""
    if len(float_array_input) == 0:
        return np.array([], dtype="float32")
    return np.concatenate([np.asarray(c, dtype="float32") for c in float_array_input])


# ---- the reference's WAV import / export ------------------------------------------------------
def _read_pcm16(wav_file_path):
    with wave.open(wav_file_path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM .wav files are supported (like the reference's int16 readers)")
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16), w.getnchannels(), w.getframerate()


def MonoWavToNumpy16BitInt(wav_file_path):
    """int16 samples of a .wav file (Utility.py:210-214)."""
    return _read_pcm16(wav_file_path)[0]


def MonoWavToNumpyFloat(wav_file_path):
    """float32 samples in [-1, 1): int16 / 32768 (Utility.py:233-238)."""
    return _read_pcm16(wav_file_path)[0].astype("float32") / 32768


def StereoWavToNumpyFloat(wav_file_path):
    """(left, right) float32 arrays of a stereo file (Utility.py:256-278)."""
    audio, n_channels, _ = _read_pcm16(wav_file_path)
    if n_channels != 2:
        raise ValueError("This function supports only stereo .wav files.")
    audio = audio.reshape(-1, 2).astype("float32") / 32768
    return audio[:, 0], audio[:, 1]


def float_to_pcm16(numpy_array):
    """(x * 32767).astype(int16): the reference's export conversion (Utility.py:306)."""
    return (np.asarray(numpy_array) * 32767).astype("int16")


def NumpyFloatToWav(wav_file_path, numpy_array):
    """Write a float array (mono (n,), stereo (n, 2) or (2, n)) as 16-bit PCM (Utility.py:295-312)."""
    numpy_array = np.asarray(numpy_array)
    if numpy_array.ndim == 2 and numpy_array.shape[0] == 2:
        numpy_array = numpy_array.T
    n_channels = 1 if numpy_array.ndim == 1 else numpy_array.shape[1]
    if not np.any((numpy_array >= -1) & (numpy_array <= 1)):
        raise ValueError("Array values should be in the range [-1.0, 1.0]")
    _write_pcm16(wav_file_path, float_to_pcm16(numpy_array), n_channels, config.sampling_rate)


def _write_pcm16(path, int_data, n_channels, rate):
    with wave.open(path, "wb") as w:
        w.setnchannels(n_channels)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.ascontiguousarray(int_data, dtype=np.int16).tobytes())


# ---- many files, one launch -------------------------------------------------------------------
# Engines of earlier process() calls, kept for the next one with the same filter, channel count and mode: creating an engine (tables,
# ring, plan set-up: ~10 ms, more than filtering a few hundred files) used to dominate WavBank.process.  A reused engine starts from
# zero history like a fresh one (reset).  At most _BANK_ENGINES live at a time, oldest closed first.
_BANK_ENGINES = 4
_bank_cache = {}


def _bank_engine(fir, channels, device, exact):
    import hashlib
    taps = np.ascontiguousarray(fir.taps, dtype=np.float64)
    key = (hashlib.blake2b(taps.tobytes(), digest_size=16).hexdigest(), len(taps), int(fir.chunk_size), int(fir.latency_chunks), int(fir.lookahead),
           int(channels), int(device), str(exact))
    eng = _bank_cache.pop(key, None)
    if eng is not None:
        eng.reset()
    elif exact == "fft":
        eng = FirEngine(fir, channels=channels, device=device, sample_format="s16_f64", optimize_for="batch")
    elif exact:
        eng = ExactFirEngine(fir, channels=channels, device=device, sample_format="s16")
    else:  # (make_engine: kernels longer than one transform - chunk 88200 - and chunk sizes that are not multiples of 4 go the same way)
        eng = make_engine(fir, channels=channels, device=device, sample_format="s16", optimize_for="batch")
    _bank_cache[key] = eng  # most recently used last
    while len(_bank_cache) > _BANK_ENGINES:
        _bank_cache.pop(next(iter(_bank_cache))).close()
    return eng


def _over_channel_groups(channels, work, threads=8):
    """work(a, b) for disjoint channel ranges on a few threads (numpy releases the GIL inside large copies)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = max(1, min(threads, os.cpu_count() or 1, channels))
    if n == 1:
        work(0, channels)
        return
    bounds = [channels * i // n for i in range(n + 1)]
    with ThreadPoolExecutor(max_workers=n) as pool:
        list(pool.map(lambda i: work(bounds[i], bounds[i + 1]), range(n)))


def close_bank_engines():
    """Release the engines WavBank.process keeps between calls."""
    while _bank_cache:
        _bank_cache.pop(next(iter(_bank_cache))).close()


class WavBank:
    """A bank of 16-bit WAV files filtered together.

    Every channel of every file becomes one mono channel of an int16 engine (a stereo file = two
    independent devices, as in Example2.py:13-21).  Files are padded with silence to the longest one,
    rounded up to whole chunks (MakeChunks' padding).  `process(fir)` runs the whole bank in one
    multi-step launch and returns int16 data per file, delayed by one chunk like the reference's
    device loop (the last input chunk is never flushed, Example1.py:16-18)."""

    def __init__(self, paths, chunk_size=None):
        self.chunk_size = int(chunk_size or config.chunk_size)
        self.files = []  # (path, n_channels, n_frames, rate)
        chans = []
        for p in paths:
            audio, n_ch, rate = _read_pcm16(p)
            frames = audio.reshape(-1, n_ch)
            self.files.append((p, n_ch, frames.shape[0], rate))
            chans.extend(frames[:, c] for c in range(n_ch))
        longest = max(f[2] for f in self.files)
        self.steps = -(-longest // self.chunk_size)
        self.pcm = np.zeros((len(chans), self.steps * self.chunk_size), np.int16)
        for i, c in enumerate(chans):
            self.pcm[i, : len(c)] = c
        self.channels = len(chans)

    def batch(self):
        """[steps, channels, chunk] int16, the engine's batch layout - built once per bank (the files do not change) by a few threads:
        a single-threaded numpy transposition of a few hundred megabytes took as long as the whole rest of process()."""
        if getattr(self, "_batch", None) is None:
            self._batch = np.empty((self.steps, self.channels, self.chunk_size), np.int16)
            src = self.pcm.reshape(self.channels, self.steps, self.chunk_size)
            _over_channel_groups(self.channels, lambda a, b: self._batch[:, a:b].__setitem__(slice(None), src[a:b].transpose(1, 0, 2)))
        return self._batch

    def process(self, fir: FirStream, device=0, exact=False):
        """exact=False: the int16 FFT engine in float32 (within one LSB of the reference's WAV output, a few samples per ten
        thousand differ because the export truncates); exact=True: the float64 direct-sum engine, which reproduces the
        reference's int16 stream bit for bit (O(taps) per sample - fine for files); exact="fft": the FFT engine in FLOAT64
        (sample_format "s16_f64") - the direct sum's int16 stream except where the float64 result lies within ~1e-15 of a
        float32 rounding boundary (about one sample in ten million), at a third of the float32 engine's rate."""
        eng = _bank_engine(fir, self.channels, device, exact)
        out = eng.apply_host(self.batch())  # [steps, C, N] int16
        flat = np.empty((self.channels, self.steps * self.chunk_size), np.int16)
        dst = flat.reshape(self.channels, self.steps, self.chunk_size)
        _over_channel_groups(self.channels, lambda a, b: dst[a:b].__setitem__(slice(None), out[:, a:b].transpose(1, 0, 2)))
        result, c0 = [], 0
        for _, n_ch, _, _ in self.files:
            result.append(flat[c0] if n_ch == 1 else flat[c0:c0 + n_ch].T)
            c0 += n_ch
        return result

    def write(self, outputs, paths):
        for data, path, (_, n_ch, _, rate) in zip(outputs, paths, self.files):
            _write_pcm16(path, data, n_ch, rate)
