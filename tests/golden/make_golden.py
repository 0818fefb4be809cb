#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (pyAudioDspTools @ /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

Writes small ``.npz`` fixtures next to this script.  Inputs are regenerated from seeds by the
tests (``numpy.random.default_rng(seed).uniform(-1, 1, n).astype(float32)``), so only outputs,
design kernels and a few spot values are stored.  Nothing from the reference's source text is
stored - only numbers it produced.
"""
import hashlib
import io
import os
import sys
import contextlib

import numpy as np

REF = os.environ.get("ADSP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):  # the reference prints a cupy info line on import
    import pyAudioDspTools as ref  # noqa: E402


def stream(seed, n_total):
    return np.random.default_rng(seed).uniform(-1, 1, n_total).astype(np.float32)


def run_device(dev, x, n):
    return np.concatenate([dev.apply(x[i * n:(i + 1) * n]) for i in range(len(x) // n)])


def taps_of(spectrum, taps):
    return np.fft.ifft(spectrum).real[:taps].copy()


ONLY = set(a for a in sys.argv[1:] if not a.startswith("-"))  # e.g. `make_golden.py kat_gate`: rewrite just that file


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)} bytes")


def main():
    meta = {"numpy": np.__version__,
            "note_kat_example1_full": "kat_example1_full.npz (make_golden_example1_full.py) carries the reference's whole 16-bit mono test WAV "
                                      "(TestFile16BitMono.wav, 264600 samples) as int16 PCM DATA - the input of the full Example1 run - next to "
                                      "outputs of the reference; no source text of the reference is stored anywhere under tests/"}

    # ---- design kernels -------------------------------------------------------------
    design = {}
    for fs, n, fc in [(44100, 4096, 800), (44100, 512, 200), (96000, 8192, 800), (44100, 4096, 160), (48000, 1024, 50)]:
        ref.config.initialize(fs, n)
        dev = ref.CreateLowCutFilter(fc)
        design[f"lowcut_{fs}_{n}_{fc}"] = taps_of(dev.sinc_filter, dev.filter_length)
    for fs, n, fc in [(44100, 4096, 8000), (44100, 512, 8000), (96000, 8192, 8000), (48000, 1024, 20000)]:
        ref.config.initialize(fs, n)
        dev = ref.CreateHighCutFilter(fc)
        design[f"highcut_{fs}_{n}_{fc}"] = taps_of(dev.sinc_filter, dev.filter_length)
    for fs, n, p in [(44100, 512, (100, 2, 700, -4, 8000, 5)), (44100, 4096, (100, 2, 700, -4, 8000, 5)),
                     (96000, 8192, (100, 2, 700, -4, 8000, 5)), (48000, 1024, (250, -6, 1500, 3, 6000, -2.5))]:
        ref.config.initialize(fs, n)
        dev = ref.CreateEQ3BandFFT(*p)
        tag = f"eq_{fs}_{n}_" + "_".join(str(v) for v in p)
        design[tag + "_highshelf"] = taps_of(dev.sinc_filter_highshelf, dev.filter_length)
        design[tag + "_lowshelf"] = taps_of(dev.sinc_filter_lowshelf, dev.filter_length)
        design[tag + "_mid_lowpass"] = taps_of(dev.sinc_filter_mid_lowpass, dev.filter_length)
        design[tag + "_mid_highpass"] = taps_of(dev.sinc_filter_mid_highpass, dev.filter_length)
    # even filter lengths (chunk sizes with N // 2 odd): the spectral inversion adds its 1 at (L - 1) // 2 = L/2 - 1
    for fs, n, fc in [(48000, 1002, 500), (44100, 30, 3000)]:
        ref.config.initialize(fs, n)
        design[f"lowcut_{fs}_{n}_{fc}"] = taps_of(ref.CreateLowCutFilter(fc).sinc_filter, n // 2 - 1)
        design[f"highcut_{fs}_{n}_{fc}"] = taps_of(ref.CreateHighCutFilter(fc).sinc_filter, n // 2 - 1)
    ref.config.initialize(48000, 1002)
    dev = ref.CreateEQ3BandFFT(250, -6, 1500, 3, 6000, -2.5)
    for part in ("highshelf", "lowshelf", "mid_lowpass", "mid_highpass"):
        design["eq_48000_1002_250_-6_1500_3_6000_-2.5_" + part] = taps_of(getattr(dev, "sinc_filter_" + part), dev.filter_length)
    # default-argument constructors (EffectFFTFilter.py:18 / :91)
    ref.config.initialize(44100, 512)
    design["highcut_default_44100_512"] = taps_of(ref.CreateHighCutFilter().sinc_filter, 255)
    design["lowcut_default_44100_512"] = taps_of(ref.CreateLowCutFilter().sinc_filter, 255)
    # spectrum spot values quoted in SURVEY 8c
    ref.config.initialize(44100, 4096)
    hc = ref.CreateHighCutFilter(8000)
    lc = ref.CreateLowCutFilter(800)
    design["spot_B_H01"] = np.array([hc.sinc_filter[0], hc.sinc_filter[1]])
    design["spot_A_H0_Hnyq"] = np.array([lc.sinc_filter[0], lc.sinc_filter[3 * 4096 // 2]])
    save("design", **design)

    # ---- KAT streams A-D (seed 1234, 6 chunks) ----------------------------------------
    kat = {}
    ref.config.initialize(44100, 4096)
    kat["A"] = run_device(ref.CreateLowCutFilter(800), stream(1234, 6 * 4096), 4096)
    kat["B"] = run_device(ref.CreateHighCutFilter(8000), stream(1234, 6 * 4096), 4096)
    ref.config.initialize(44100, 512)
    kat["C"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(1234, 6 * 512), 512)
    kat["D"] = run_device(ref.CreateLowCutFilter(200), stream(1234, 6 * 512), 512)
    # extra shapes: EQ at 4096, filters at 8192/96k, N=1024/48k, N=64 (smallest supported)
    ref.config.initialize(44100, 4096)
    kat["EQ4096"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(77, 5 * 4096), 4096)
    ref.config.initialize(96000, 8192)
    kat["LC8192"] = run_device(ref.CreateLowCutFilter(800), stream(78, 4 * 8192), 8192)
    kat["EQ8192"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(79, 4 * 8192), 8192)
    ref.config.initialize(48000, 1024)
    kat["HC1024"] = run_device(ref.CreateHighCutFilter(20000), stream(80, 7 * 1024), 1024)
    kat["EQ1024"] = run_device(ref.CreateEQ3BandFFT(250, -6, 1500, 3, 6000, -2.5), stream(81, 7 * 1024), 1024)
    ref.config.initialize(44100, 2048)
    kat["LC2048"] = run_device(ref.CreateLowCutFilter(160), stream(82, 5 * 2048), 2048)
    ref.config.initialize(44100, 256)
    kat["HC256"] = run_device(ref.CreateHighCutFilter(3000), stream(83, 9 * 256), 256)
    ref.config.initialize(44100, 128)
    kat["EQ128"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(84, 9 * 128), 128)
    ref.config.initialize(44100, 64)
    kat["LC64"] = run_device(ref.CreateLowCutFilter(2000), stream(85, 11 * 64), 64)
    # chunk sizes that are not powers of two (any N % 4 == 0 works in the reference)
    ref.config.initialize(44100, 1000)
    kat["LC1000"] = run_device(ref.CreateLowCutFilter(300), stream(86, 7 * 1000), 1000)
    kat["EQ1000"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(87, 7 * 1000), 1000)
    ref.config.initialize(48000, 1920)
    kat["HC1920"] = run_device(ref.CreateHighCutFilter(9000), stream(88, 5 * 1920), 1920)
    ref.config.initialize(44100, 12000)
    kat["LC12000"] = run_device(ref.CreateLowCutFilter(120), stream(89, 3 * 12000), 12000)
    ref.config.initialize(44100, 20)
    kat["EQ20"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(90, 40 * 20), 20)
    # chunk sizes that are NOT multiples of 4 (round 4): the reference takes any chunk_size; N // 2 odd gives an
    # EVEN filter length (N = 30: 14 taps, N = 1002: 500), whose slice look-ahead L // 2 is one more than (L - 1) // 2
    ref.config.initialize(44100, 30)
    kat["LC30"] = run_device(ref.CreateLowCutFilter(3000), stream(93, 30 * 30), 30)
    kat["HC30"] = run_device(ref.CreateHighCutFilter(8000), stream(94, 30 * 30), 30)
    kat["EQ30"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(95, 30 * 30), 30)
    ref.config.initialize(44100, 1001)
    kat["LC1001"] = run_device(ref.CreateLowCutFilter(300), stream(96, 7 * 1001), 1001)
    kat["EQ1001"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(97, 7 * 1001), 1001)
    ref.config.initialize(48000, 1002)
    kat["LC1002"] = run_device(ref.CreateLowCutFilter(500), stream(98, 7 * 1002), 1002)
    kat["HC1002"] = run_device(ref.CreateHighCutFilter(9000), stream(99, 7 * 1002), 1002)
    kat["EQ1002"] = run_device(ref.CreateEQ3BandFFT(250, -6, 1500, 3, 6000, -2.5), stream(100, 7 * 1002), 1002)
    ref.config.initialize(44100, 6)   # two taps
    kat["HC6"] = run_device(ref.CreateHighCutFilter(8000), stream(101, 50 * 6), 6)
    ref.config.initialize(44100, 4410)  # 100 ms at 44.1 kHz: N % 4 == 2, 2204 taps
    kat["LC4410"] = run_device(ref.CreateLowCutFilter(160), stream(102, 4 * 4410), 4410)
    # Example4's chunk size (88200 = 2 s at 44.1 kHz, 44099 taps): every 64th output sample of 3 chunks
    ref.config.initialize(44100, 88200)
    kat["LC88200_dec64"] = run_device(ref.CreateLowCutFilter(300), stream(91, 3 * 88200), 88200)[::64].copy()
    kat["EQ88200_dec64"] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), stream(92, 3 * 88200), 88200)[::64].copy()
    for k in "ABCD":
        meta["sha_" + k] = hashlib.sha256(kat[k].tobytes()).hexdigest()[:12]
    save("kat_streams", **kat)

    # ---- E: chain LowCut(800) -> EQ3 -> HighCut(8000), 96 kHz, N=8192, 12 chunks ------------
    ref.config.initialize(96000, 8192)
    a, b, c = ref.CreateLowCutFilter(800), ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), ref.CreateHighCutFilter(8000)
    x = stream(4321, 12 * 8192)
    y = np.concatenate([c.apply(b.apply(a.apply(x[i * 8192:(i + 1) * 8192]))) for i in range(12)])
    save("kat_chain", E=y)

    # ---- F: Example1 plumbing on the reference's own 16-bit mono WAV -------------------------
    ref.config.initialize(44100, 4096)
    with contextlib.redirect_stdout(io.StringIO()):
        full = ref.Utility.MonoWavToNumpyFloat(os.path.join(REF, "TestFile16BitMono.wav"))
    chunks = ref.MakeChunks(full)
    dev = ref.CreateLowCutFilter(800)
    outs = [dev.apply(ch) for ch in chunks]
    merged = ref.CombineChunks(outs)
    meta["example1_len_in"] = int(len(full))
    meta["example1_n_chunks"] = int(len(chunks))
    meta["example1_sha16"] = hashlib.sha256(merged.tobytes()).hexdigest()[:16]
    first8_in = np.round(np.concatenate(chunks[:8]) * 32768).astype(np.int16)  # exact: samples are int16/32768
    assert np.array_equal(first8_in.astype(np.float32) / 32768, np.concatenate(chunks[:8]))
    save("kat_example1", pcm16_first8=first8_in, out_first8=np.concatenate(outs[:8]),
         out_len=np.array([len(merged)]))

    # ---- F2: Example2 plumbing (stereo = two independent devices) on the reference's 16-bit stereo WAV ----
    ref.config.initialize(44100, 4096)
    with contextlib.redirect_stdout(io.StringIO()):
        left, right = ref.Utility.StereoWavToNumpyFloat(os.path.join(REF, "TestFile16BitStereo.wav"))
    cl, cr = ref.MakeChunks(left), ref.MakeChunks(right)
    dl, dr = ref.CreateLowCutFilter(800), ref.CreateLowCutFilter(800)
    ol = [dl.apply(c) for c in cl[:4]]
    orr = [dr.apply(c) for c in cr[:4]]
    pcm = np.stack([np.round(np.concatenate(cl[:4]) * 32768), np.round(np.concatenate(cr[:4]) * 32768)], axis=1).astype(np.int16)
    save("kat_example2", pcm16_first4_stereo=pcm, out_left=np.concatenate(ol), out_right=np.concatenate(orr),
         n_frames=np.array([len(left)]))

    # ---- G: edge inputs, N=512 for each device -----------------------------------------------
    n = 512
    edge_inputs = {
        "zeros": np.zeros(5 * n, np.float32),
        "imp0": np.eye(1, 5 * n, 0, dtype=np.float32)[0],
        "impNm1": np.eye(1, 5 * n, n - 1, dtype=np.float32)[0],
        "impN": np.eye(1, 5 * n, n, dtype=np.float32)[0],
        "dc": np.ones(5 * n, np.float32),
        "square": np.where((np.arange(5 * n) // 37) % 2 == 0, 1.0, -1.0).astype(np.float32),
    }
    edge = {}
    for name, x in edge_inputs.items():
        ref.config.initialize(44100, n)
        edge["lowcut_" + name] = run_device(ref.CreateLowCutFilter(200), x, n)
        edge["highcut_" + name] = run_device(ref.CreateHighCutFilter(8000), x, n)
        edge["eq_" + name] = run_device(ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5), x, n)
    # input-type behaviour: float64 chunks and python lists are accepted (concatenate axis=None)
    ref.config.initialize(44100, n)
    x64 = np.random.default_rng(5).uniform(-1, 1, 4 * n)
    edge["lowcut_f64in"] = run_device(ref.CreateLowCutFilter(200), x64, n)
    dev = ref.CreateHighCutFilter(8000)
    edge["highcut_listin"] = np.concatenate([dev.apply(list(x64[i * n:(i + 1) * n])) for i in range(4)])
    save("kat_edges", **edge)

    # ---- H: stateless wave-shapers, alone and behind an FFT device ----------------
    fx = {}
    loud = (stream(100, 4096) * np.float32(1.5)).astype(np.float32)  # |x| up to 1.5: exercises every clipping branch
    fx["softclip_044"] = ref.CreateSoftClipper().apply(loud)
    fx["softclip_200"] = ref.CreateSoftClipper(2.0).apply(loud)
    fx["harddist"] = ref.CreateHardDistortion().apply(loud)
    fx["saturator_hard"] = ref.CreateSaturator().apply(loud)
    fx["saturator_soft"] = ref.CreateSaturator(-12.0, 3.0, 'soft').apply(loud)
    fx["volume_p6_clip"] = ref.VolumeChange(loud, 6.0)
    fx["volume_m35_noclip"] = ref.VolumeChange(loud, -3.5, False)
    from pyAudioDspTools import _EffectBitCrusher as ref_crusher
    fx["bitcrusher"] = ref_crusher.CreateBitCrusher().apply(stream(100, 4096))  # |x| <= 1: the int16 cast does not wrap
    ref.config.initialize(44100, 512)
    x = stream(101, 8 * 512)
    for tag, make in [("lowcut_softclip", lambda: (ref.CreateLowCutFilter(200), ref.CreateSoftClipper(0.44))),
                      ("highcut_harddist", lambda: (ref.CreateHighCutFilter(8000), ref.CreateHardDistortion())),
                      ("eq_saturator_soft", lambda: (ref.CreateEQ3BandFFT(100, 2, 700, -4, 8000, 5),
                                                     ref.CreateSaturator(-12.0, 3.0, 'soft')))]:
        dev, eff = make()
        fx["chain512_" + tag] = np.concatenate([eff.apply(dev.apply(x[i * 512:(i + 1) * 512])) for i in range(8)])
    ref.config.initialize(44100, 4096)
    x = stream(102, 5 * 4096)
    dev, eff = ref.CreateLowCutFilter(800), ref.CreateSaturator()
    fx["chain4096_lowcut_saturator_hard"] = np.concatenate([eff.apply(dev.apply(x[i * 4096:(i + 1) * 4096])) for i in range(5)])
    dev = ref.CreateEQ3BandFFT(100, 6, 700, 3, 8000, 6)  # boosts: output exceeds 1.0, the clip matters
    fx["chain4096_eq_volume_p3"] = np.concatenate([ref.VolumeChange(dev.apply(x[i * 4096:(i + 1) * 4096]), 3.0) for i in range(5)])
    # tremolo: default (period 9800 samples), a non-integer period (48000 / 7 -> 6858-sample table), and the buffer
    # quirk (period 1536 = 3 chunks of 512: the third chunk empties the buffer "exactly" and is replayed for ever)
    ref.config.initialize(44100, 4096)
    x = stream(103, 6 * 4096)
    t = ref.CreateTremolo()
    fx["tremolo_default"] = np.concatenate([t.apply(x[i * 4096:(i + 1) * 4096]) for i in range(6)])
    ref.config.initialize(48000, 1000)
    x = stream(104, 12 * 1000)
    t = ref.CreateTremolo(0.9, 7)
    fx["tremolo_48k_7hz"] = np.concatenate([t.apply(x[i * 1000:(i + 1) * 1000]) for i in range(12)])
    fx["tremolo_48k_7hz_len"] = np.array([t.lfo_length])
    ref.config.initialize(44100, 512)
    x = stream(105, 8 * 512)
    t = ref.CreateTremolo(0.5, 44100 / 1536)
    fx["tremolo_quirk"] = np.concatenate([t.apply(x[i * 512:(i + 1) * 512]) for i in range(8)])
    dev, t = ref.CreateLowCutFilter(200), ref.CreateTremolo(0.6, 10)
    x = stream(106, 12 * 512)
    fx["chain512_lowcut_tremolo"] = np.concatenate([t.apply(dev.apply(x[i * 512:(i + 1) * 512])) for i in range(12)])
    dev, t = ref.CreateHighCutFilter(8000), ref.CreateTremolo(0.5, 44100 / 1536)
    fx["chain512_highcut_tremolo_quirk"] = np.concatenate([t.apply(dev.apply(x[i * 512:(i + 1) * 512])) for i in range(12)])
    # MixSignals: three loud signals (clipping), and three FFT devices mixed per chunk
    a, b, c = stream(107, 4096), stream(108, 4096), stream(109, 4096)
    fx["mix3"] = ref.MixSignals(a, b, c)
    d1, d2, d3 = ref.CreateLowCutFilter(200), ref.CreateHighCutFilter(8000), ref.CreateEQ3BandFFT(100, 6, 700, 3, 8000, 6)
    xs = [stream(110 + k, 6 * 512) for k in range(3)]
    fx["chain512_mix3"] = np.concatenate([ref.MixSignals(d1.apply(xs[0][i * 512:(i + 1) * 512]), d2.apply(xs[1][i * 512:(i + 1) * 512]),
                                                         d3.apply(xs[2][i * 512:(i + 1) * 512])) for i in range(6)])
    save("kat_effects", **fx)

    # ---- I: callers that embed the FFT filters: CreateDelay and the private reverb -------------
    dl = {}
    ref.config.initialize(44100, 4096)
    x = stream(130, 14 * 4096)
    for tag, kw in [("default", {}), ("wet", {"wet": True}), ("300ms_4loops", {"time_in_ms": 300, "feedback_loops": 4})]:
        d = ref.CreateDelay(**kw)
        dl["delay4096_" + tag] = np.concatenate([d.apply(x[i * 4096:(i + 1) * 4096].copy()) for i in range(14)])
    ref.config.initialize(44100, 512)
    x = stream(131, 12 * 512)
    d = ref.CreateDelay(10, 5)      # 441-sample taps: several land inside one chunk
    dl["delay512_10ms_5loops"] = np.concatenate([d.apply(x[i * 512:(i + 1) * 512].copy()) for i in range(12)])
    d = ref.CreateDelay(7.3, 1, wet=True)   # 321 samples, one tap
    dl["delay512_7ms_wet"] = np.concatenate([d.apply(x[i * 512:(i + 1) * 512].copy()) for i in range(12)])
    d = ref.CreateDelay(100, 0)     # no taps at all
    dl["delay512_noloops"] = np.concatenate([d.apply(x[i * 512:(i + 1) * 512].copy()) for i in range(3)])
    from pyAudioDspTools import _EffectReverb as ref_reverb
    with contextlib.redirect_stdout(io.StringIO()):  # the delay lines print their tap spacing
        rv = ref_reverb.CreateReverb()
    x = stream(132, 40 * 512)
    dl["reverb512_default"] = np.concatenate([rv.applyreverb(x[i * 512:(i + 1) * 512].copy()) for i in range(40)])
    ref.config.initialize(48000, 256)
    with contextlib.redirect_stdout(io.StringIO()):
        rv = ref_reverb.CreateReverb(800)
    x = stream(133, 60 * 256)
    dl["reverb256_800ms_48k"] = np.concatenate([rv.applyreverb(x[i * 256:(i + 1) * 256].copy()) for i in range(60)])
    save("kat_callers", **dl)

    # ---- J: the recursive devices (last item): IIR 3-band EQ and the compressor --------------------
    rc = {}
    ref.config.initialize(44100, 1024)
    x = stream(160, 6 * 1024)
    eq = ref.CreateEQ3Band(100, 2, 700, -4, 8000, 5)
    rc["iir_low"] = np.concatenate([eq.applylowband(x[i * 1024:(i + 1) * 1024].copy()) for i in range(6)])
    rc["iir_mid"] = np.concatenate([eq.applymidband(x[i * 1024:(i + 1) * 1024].copy()) for i in range(6)])
    rc["iir_high"] = np.concatenate([eq.applyhighband(x[i * 1024:(i + 1) * 1024].copy()) for i in range(6)])
    eq = ref.CreateEQ3Band(250, -6, 1500, 3, 6000, -2.5)
    rc["iir_cascade"] = np.concatenate([eq.applyhighband(eq.applymidband(eq.applylowband(x[i * 1024:(i + 1) * 1024].copy())))
                                        for i in range(6)])
    rc["iir_coeffs"] = np.array([eq.LOWb0, eq.LOWb1, eq.LOWb2, eq.LOWa0, eq.LOWa1, eq.LOWa2, eq.MIDb0, eq.MIDb1, eq.MIDb2, eq.MIDa0,
                                 eq.MIDa1, eq.MIDa2, eq.HIGHb0, eq.HIGHb1, eq.HIGHb2, eq.HIGHa0, eq.HIGHa1, eq.HIGHa2])
    # compressor: bursts around the threshold exercise attack / hold / release / re-trigger / chunk boundaries
    rng = np.random.default_rng(161)
    n = 1024
    env = np.repeat(rng.choice([0.05, 0.12, 0.3, 0.9], size=12 * n // 64), 64).astype(np.float32)
    xc = (rng.uniform(-1, 1, 12 * n).astype(np.float32) * env).astype(np.float32)
    rc["comp_input"] = xc
    for tag, kw in [("default", {}), ("fast", {"threshold_in_db": -20, "ratio": 0.3, "attack_in_ms": 0.5, "release_in_ms": 2.0}),
                    ("slow", {"threshold_in_db": -10, "ratio": 0.8, "attack_in_ms": 10.0, "release_in_ms": 100.0})]:
        cp = ref.CreateCompressor(**kw)
        rc["comp_" + tag] = np.concatenate([cp.apply(xc[i * n:(i + 1) * n].copy()) for i in range(12)])
    save("kat_recursive", **rc)

    # ---- K: the gate (EffectGate.py) - same state machine as the compressor on a depth-scaled copy, threshold on the raw
    # input, sampling rate hard-wired to 44100.  Bursts around the three thresholds; the default release (8824 samples)
    # spans many chunks.  The reference leaves its argument alone (it works on `input * depth`).
    gt = {}
    rng = np.random.default_rng(162)
    n = 1024
    env = np.repeat(rng.choice([0.04, 0.2, 0.5, 1.0], size=16 * n // 128), 128).astype(np.float32)
    xg = (rng.uniform(-1, 1, 16 * n).astype(np.float32) * env).astype(np.float32)
    gt["gate_input"] = xg
    ref.config.initialize(48000, n)  # the gate ignores config: its envelopes are built for 44100 Hz whatever this says
    for tag, kw in [("default", {}), ("fast", {"threshold_in_db": -12, "depth": 0.25, "attack": 0.5, "release": 3.0}),
                    ("deep", {"threshold_in_db": -20, "depth": 0.01, "attack": 10.0, "release": 50.0})]:
        g = ref.CreateGate(**kw)
        keep = xg.copy()
        gt["gate_" + tag] = np.concatenate([g.apply(xg[i * n:(i + 1) * n]) for i in range(16)])
        assert np.array_equal(keep, xg) and gt["gate_" + tag].dtype == np.float32
    save("kat_gate", **gt)

    with open(os.path.join(HERE, "META.txt"), "w") as fh:
        for k in sorted(meta):
            fh.write(f"{k} = {meta[k]}\n")
    print(meta)


if __name__ == "__main__":
    main()
