"""Counter-based synthetic input: sample (channel c, absolute index t) is a pure function of (seed, c, t).

``fill_device`` runs the HIP generator (csrc/adsp_synth.hip, C ABI ``adsp_synth_device``) straight into a resident
``[steps, channels, chunk]`` batch; ``uniform_host`` / ``pcm16_host`` are its numpy twins, bit-identical, so any channel of a
resident batch can be regenerated on the host - which is how bench.py checks the output of its timed region against the CPU
oracle without copying the batch back.
"""
import ctypes

import numpy as np

from . import _capi

_M32 = np.uint64(0xFFFFFFFF)


def _hash(seed, channel, t):
    """32-bit hash of (seed, channel, absolute sample index t) - integer arithmetic mod 2^32, then MurmurHash3's finaliser."""
    t = np.asarray(t, dtype=np.uint64)
    lo, hi = t & _M32, t >> np.uint64(32)
    h = (lo * np.uint64(0x9E3779B1) + hi * np.uint64(0x85EBCA77) + np.uint64(int(channel) & 0xFFFFFFFF) * np.uint64(0xC2B2AE3D)
         + np.uint64(int(seed) & 0xFFFFFFFF) * np.uint64(0x27D4EB2F)) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def uniform_host(seed, channel, first_sample, count, amplitude=1.0):
    """float32[count]: samples first_sample .. first_sample + count - 1 of `channel` - what fill_device writes there."""
    h = _hash(seed, channel, np.uint64(first_sample) + np.arange(count, dtype=np.uint64))
    v = (h >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    return v * np.float32(amplitude)


def pcm16_host(seed, channel, first_sample, count):
    """int16[count]: the int16 form (uniform in [-16384, 16384), -6 dBFS)."""
    h = _hash(seed, channel, np.uint64(first_sample) + np.arange(count, dtype=np.uint64))
    return ((h >> np.uint64(17)).astype(np.int64) - 16384).astype(np.int16)


def batch_host(seed, first_channel, n_channels, first_sample, chunk_size, n_steps, sample_format="f32", amplitude=1.0):
    """[n_steps, n_channels, chunk_size]: the whole batch fill_device would write (tests; small shapes)."""
    gen = (lambda c: uniform_host(seed, c, first_sample, n_steps * chunk_size, amplitude)) if sample_format == "f32" else \
          (lambda c: pcm16_host(seed, c, first_sample, n_steps * chunk_size))
    rows = [gen(first_channel + c).reshape(n_steps, chunk_size) for c in range(n_channels)]
    return np.ascontiguousarray(np.stack(rows, axis=1))


def fill_device(d_out, seed, first_channel, first_sample, n_channels, chunk_size, n_steps, sample_format="f32", amplitude=1.0,
                device=0, stream=None):
    """Fill the device batch `d_out` ([n_steps, n_channels, chunk_size], float32 or int16; a torch tensor or an address)."""
    from .engine import _ptr
    fmt = _capi.ADSP_FORMAT_F32 if sample_format == "f32" else _capi.ADSP_FORMAT_S16
    _capi.check(_capi.load().adsp_synth_device(int(device), ctypes.c_uint(int(seed) & 0xFFFFFFFF), ctypes.c_uint(int(first_channel) & 0xFFFFFFFF),
                                               ctypes.c_ulonglong(int(first_sample)), int(n_channels), int(chunk_size), int(n_steps), fmt,
                                               float(amplitude), _ptr(d_out), _ptr(stream)))
