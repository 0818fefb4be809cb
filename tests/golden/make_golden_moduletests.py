#!/usr/bin/env python3
"""Golden vectors for the reference's OWN test harness, ModuleTests.py - the one script its repository holds that drives every device.
Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_moduletests.py

ModuleTests.py:34-52 initialises the package at 44100 Hz / 512 samples, :57-61 builds a 1 kHz sine, a 1 kHz square and white noise,
:67 chunks a copy of the sine, :73-84 creates one device of every kind with its DEFAULT arguments (the FFT filters and EQs with the values
of :81-84), and :95-214 sends the SAME list of chunks through ten of them one after the other, every loop overwriting the list entries
with the device's outputs (`chunks[i] = dev.apply(chunks[i])`): Saturator -> Compressor -> Delay -> Tremolo -> HardDistortion -> Gate ->
LowCut(200) -> HighCut(8000) -> EQ3BandFFT(100, 2, 700, -4, 8000, 5) -> SoftClipper, then CombineChunks (:217).

kat_moduletests.npz holds what the reference computes in that harness on a SHORTER signal (the script's own is a minute long; its
Compressor and Gate are per-sample Python loops): `stage_00` = the chunked sine, `stage_01` .. `stage_10` = the combined chunks after
each loop (so stage k is both the reference's output of device k and the input it handed device k + 1), the names of the stages, the
generators' first samples for the same arguments, the utility helpers' values on the sine (ConvertdBVTo16Bit, Convert16BitTodBV,
InfodBV, InfodBV16Bit, VolumeChange) and the band edges / level of CreateWhitenoise (its phases are unseeded random numbers: only its
magnitude spectrum is a fixed quantity).  Only numbers the reference produced are stored.

The file also holds the reference's GPU harness, ModuleTestsGPU.py: 44100 Hz / 88200 samples (:35), the chunked sine as ONE 2-D device array
(:58, `cupy.array(MakeChunks(...))`), LowCutGPU(200) -> HighCutGPU(8000) -> EQ3BandFFTGPU(100, 2, 700, -4, 8000, 5), each loop assigning
`arr[i] = dev.apply(arr[i])` (:78-110) - Example4's in-place pattern three times over.  cupy is not installed here; the twins' apply() is
the numpy classes' statement for statement (EffectFFTFilterGPU.py:66-78 = EffectFFTFilter.py:63-75), so the numpy classes run the harness on
a numpy 2-D array.  Stored for 4 chunks, every 16th sample: `gpu_inplace_k` = the array after loop k (what the script computes),
`gpu_clean_k` = the same devices fed from separate arrays (the filtered stream; what the in-place rows would hold without the aliasing).
"""
import contextlib
import copy
import io
import os
import sys

import numpy as np

REF = os.environ.get("ADSP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import pyAudioDspTools as ref  # noqa: E402
    from pyAudioDspTools._EffectReverb import CreateReverb  # noqa: E402  (ModuleTests.py:45; created at :80, never applied)

FS, N = 44100, 512           # ModuleTests.py:34
LENGTH = 30000               # the script: 44100 * 60.  30000 samples = 59 chunks: long enough for the delay's first echo (500 ms)
STAGES = ["Saturator", "Compressor", "Delay", "Tremolo", "HardDistortion", "Gate", "LowCutFilter(200)", "HighCutFilter(8000)",
          "EQ3BandFFT(100,2,700,-4,8000,5)", "SoftClipper"]


def main():
    kat = {}
    with contextlib.redirect_stdout(io.StringIO()):
        ref.config.initialize(FS, N)
        sine_full = ref.CreateSinewave(1000, LENGTH)
        square_full = ref.CreateSquarewave(1000, LENGTH)
        noise_full = ref.CreateWhitenoise(LENGTH)
        chunks = ref.MakeChunks(copy.deepcopy(sine_full))
        devices = [ref.CreateSaturator(), ref.CreateCompressor(), ref.CreateDelay(), ref.CreateTremolo(), ref.CreateHardDistortion(),
                   ref.CreateGate(), ref.CreateLowCutFilter(200), ref.CreateHighCutFilter(8000),
                   ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), ref.CreateSoftClipper()]
        CreateReverb()
        ref.CreateEQ3Band(100, 2, 700, -4, 8000, 5)
        kat["stage_00"] = np.concatenate(chunks).astype(np.float32)
        for k, dev in enumerate(devices, start=1):
            # the compressor and the gate work on the array they are handed (EffectCompressor.py:70-125): the copy keeps stage k - 1 as stored
            for i in range(len(chunks)):
                chunks[i] = dev.apply(np.array(chunks[i], copy=True))
            out = np.concatenate([np.asarray(c) for c in chunks])
            kat[f"stage_{k:02d}_dtype"] = np.array(str(out.dtype))
            kat[f"stage_{k:02d}"] = out.astype(np.float32) if out.dtype != np.float32 else out
        combined = ref.CombineChunks(chunks)
    assert np.array_equal(np.asarray(combined, dtype=np.float32), kat["stage_10"])
    kat["stage_names"] = np.array(STAGES)
    kat["length"] = np.array(LENGTH)
    kat["sine"] = sine_full
    kat["square_dtype"] = np.array(str(square_full.dtype))
    kat["square"] = square_full.astype(np.int8)
    # white noise: flat unit magnitude between 20 Hz and 20 kHz, nothing outside, scaled by 5 / n
    spec = np.abs(np.fft.rfft(noise_full.astype(np.float64))) * (1.0 / 5.0)
    freqs = np.fft.rfftfreq(LENGTH, 1.0 / FS)
    inband = (freqs >= 20) & (freqs <= 20000)
    kat["noise_dtype"] = np.array(str(noise_full.dtype))
    kat["noise_inband_bins"] = np.array([int(np.flatnonzero(inband)[0]), int(np.flatnonzero(inband)[-1])])
    kat["noise_inband_mag_minmax"] = np.array([spec[inband].min(), spec[inband].max()])
    kat["noise_outband_mag_max"] = np.array(spec[~inband].max())
    kat["noise_rms"] = np.array(float(np.sqrt(np.mean(noise_full.astype(np.float64) ** 2))))
    # utility helpers on the sine
    as16 = ref.ConvertdBVTo16Bit(sine_full * 1.5)       # clips
    kat["to16"] = as16
    kat["to16_dtype"] = np.array(str(as16.dtype))
    back = ref.Convert16BitTodBV(as16)
    kat["from16"] = back
    kat["from16_dtype"] = np.array(str(back.dtype))
    kat["info_dbv"] = np.array(ref.InfodBV(sine_full))
    kat["info_db16"] = np.array(ref.InfodBV16Bit(as16))
    vol = ref.VolumeChange(sine_full, 3.0)
    kat["volume_p3db"] = vol.astype(np.float32)
    kat["volume_dtype"] = np.array(str(vol.dtype))
    rng = np.random.default_rng(77)
    i16 = rng.integers(-32768, 32768, 4096).astype(np.int16)
    i32 = rng.integers(-2 ** 31, 2 ** 31, 4096).astype(np.int64)
    d8 = ref.Dither16BitTo8Bit(i16)
    d16 = ref.Dither32BitIntTo16BitInt(i32)
    # the dither is an unseeded draw from {-1, 0}: what is fixed is the undithered value and that the result lies 0 or 1 below it
    kat["dither8_dtype"] = np.array(str(d8.dtype))
    kat["dither16_dtype"] = np.array(str(d16.dtype))
    kat["dither8_offsets"] = np.unique(np.clip(np.around(i16 / 256), -127, 127) - d8)
    kat["dither16_offsets"] = np.unique(np.clip(np.around(i32 / 65535), -32767, 32767) - d16)
    gpu_harness(kat)
    np.savez_compressed(os.path.join(HERE, "kat_moduletests.npz"), **kat)
    for k in sorted(kat):
        v = kat[k]
        print(k, v.dtype, v.shape, (float(np.abs(v).max()) if v.dtype.kind in "fi" and v.size else v))


GPU_N, GPU_CHUNKS, GPU_DEC = 88200, 4, 16    # ModuleTestsGPU.py:35; the script runs 30 chunks


def gpu_harness(kat):
    with contextlib.redirect_stdout(io.StringIO()):
        ref.config.initialize(FS, GPU_N, use_gpu=True)
        sine = ref.CreateSinewave(1000, GPU_N * GPU_CHUNKS)
        make = [lambda: ref.CreateLowCutFilter(200), lambda: ref.CreateHighCutFilter(8000),
                lambda: ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5)]
        arr = np.array(ref.MakeChunks(copy.deepcopy(sine)))      # one 2-D array, rows overwritten in place
        clean = [np.array(c) for c in ref.MakeChunks(copy.deepcopy(sine))]
        assert arr.dtype == np.float32 and arr.shape == (GPU_CHUNKS, GPU_N)
        for k, mk in enumerate(make, start=1):
            dev = mk()
            for i in range(len(arr)):
                arr[i] = dev.apply(arr[i])
            kat[f"gpu_inplace_{k}"] = arr.reshape(-1)[::GPU_DEC].copy()
            dev = mk()
            clean = [dev.apply(c) for c in clean]
            kat[f"gpu_clean_{k}"] = np.concatenate(clean).astype(np.float32)[::GPU_DEC]
        ref.config.initialize(FS, N)
    kat["gpu_shape"] = np.array([GPU_CHUNKS, GPU_N, GPU_DEC])


if __name__ == "__main__":
    main()
