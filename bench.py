#!/usr/bin/env python3
"""bench.py - throughput of the batched FFT filter hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): CreateLowCutFilter(800) @ 44.1 kHz on 4096 mono channels x
4096-sample chunks per GPU, float32, synthetic uniform(-1,1) input already resident in HBM.

A "step" filters one [channels, chunk] batch (one chunk per channel = one reference apply() per
device).  Default mode "batch": the K steps' inputs sit in HBM as [steps, channels, chunk] arrays
and are handed to adsp_apply_device `--steps-per-launch` steps at a time (many chunks of every
channel batched as one grid; each 2N transform then keeps 1.5 N samples).  `--mode stream` runs one
launch per step through the zero-copy ring (adsp_apply_ring), the real-time call pattern; its
figure is also measured (after the timed region) and reported under "stream" in the same line.
History is carried by the engine exactly as between reference apply() calls; every output sample of
every step is produced inside the timed region.  Before the W warmup steps the same workload runs untimed for
--prewarm-ms (default 300 ms): an idle MI355X needs tens of milliseconds of sustained load before its shader clock
has ramped up, and a measurement that starts earlier reports the ramp, not the kernel.

For N > 1 (torchrun, one rank per GPU) every rank owns its own channel shard (weak scaling); the
only collective is the RCCL broadcast of the filter spectrum before the timed region.

Prints ONE JSON line on rank 0 (driver contract) including
  roofline     - algorithmic bytes (8 B/sample) / average KERNEL duration (HIP events around each
                 kernel launch on the launch stream, adsp_kernel_time) vs the 8 TB/s HBM3E spec
  cpu_baseline - the oracle's literal restatement of the reference (numpy, 1 core), bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
ALG_BYTES_PER_SAMPLE = 8  # 4 B read + 4 B written, SURVEY.md 8(d)  (float32; int16 PCM batches: 2 + 2)

FILTER_NAMES = {"lowcut": "CreateLowCutFilter(800)", "highcut": "CreateHighCutFilter(8000)",
                "eq3": "CreateEQ3BandFFT(100,2,700,-4,8000,5)",
                "chain": "LowCut(800)->EQ3BandFFT(100,2,700,-4,8000,5)->HighCut(8000) fused"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1536)
    ap.add_argument("--warmup", type=int, default=768)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed run of the same workload before the W warmup steps, until this much wall time has passed: "
                         "the shader clock of an idle MI355X takes tens of milliseconds of sustained load to ramp up "
                         "(0.38 -> 0.47 of the roofline between a 1 ms and a 100 ms warm-up)")
    ap.add_argument("--channels", type=int, default=4096, help="channels PER GPU")
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--fs", type=int, default=44100)
    ap.add_argument("--filter", default="lowcut", choices=sorted(FILTER_NAMES))
    ap.add_argument("--mode", default="batch", choices=["batch", "offline", "stream"],
                    help="batch (= offline): --steps-per-launch steps per launch; stream: one launch per step, zero-copy ring")
    ap.add_argument("--steps-per-launch", type=int, default=96,
                    help="batch mode: chunks per channel per launch.  Multiples of 3 tile exactly (3 chunks = 2 blocks of 1.5 N kept samples)")
    ap.add_argument("--ring-slots", type=int, default=4, help="stream mode: input ring length (the library default, 2 x history; fewer slots stay in the 256 MB Infinity Cache: 48 vs 51 us per step at 8)")
    ap.add_argument("--fft-mult", type=int, default=0, help="force transform length = this multiple of the chunk (0 = smallest)")
    ap.add_argument("--io", default="f32", choices=["f32", "s16"],
                    help="sample format of the resident batches: float32 (headline) or int16 PCM (fused WAV front end, 4 B/sample)")
    ap.add_argument("--effect", default="none", choices=["none", "softclip", "harddist", "saturator", "volume", "tremolo"],
                    help="fuse a stateless wave-shaper on the kernel's output (not part of the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()
    if a.mode == "offline":
        a.mode = "batch"
    return a


def make_fir(args):
    from pyaudiodsptools_amd import design
    n, fs = args.chunk, args.fs
    lc = design.FirStream(design.lowcut_kernel(800, fs, n), n)
    hc = design.FirStream(design.highcut_kernel(8000, fs, n), n)
    eq = design.FirStream(design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n), n)
    return {"lowcut": lc, "highcut": hc, "eq3": eq, "chain": lc.then(eq).then(hc)}[args.filter]


def cpu_baseline(args):
    """ModuleTests.py:168-178 style timing of the reference's algorithm (oracle port), one core."""
    from oracle import fftfilter_oracle as orc
    n, fs = args.chunk, args.fs
    if args.filter == "lowcut":
        dev = orc.OracleLowCut(800, fs, n)
    elif args.filter == "highcut":
        dev = orc.OracleHighCut(8000, fs, n)
    elif args.filter == "eq3":
        dev = orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n)
    else:
        a, b, c = (orc.OracleLowCut(800, fs, n), orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n),
                   orc.OracleHighCut(8000, fs, n))

        class _Chain:
            def apply(self, x):
                return c.apply(b.apply(a.apply(x)))
        dev = _Chain()
    rng = np.random.default_rng(1234)
    chunks = [rng.uniform(-1, 1, n).astype(np.float32) for _ in range(64)]
    for ch in chunks[:8]:
        dev.apply(ch)
    done = 0
    t0 = time.perf_counter()
    while True:
        for ch in chunks:
            dev.apply(ch)
        done += len(chunks)
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds:
            break
    return {"value": round(done * n / el / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"1 channel x {done} chunks of {n} samples, numpy {np.__version__} literal 3N complex fft/ifft "
                      f"(oracle restatement of the reference's apply), {el:.1f} s on 1 of {os.cpu_count()} host cores"}


class Runner:
    """One measured configuration: engine + resident synthetic data + a run(k_steps) closure."""

    def __init__(self, args, mode, fir, dev, local_rank, world, rank):
        import torch
        from pyaudiodsptools_amd import dist as adist
        self.torch = torch
        self.mode = mode
        C, N = args.channels, args.chunk
        self.C, self.N = C, N
        stream_mode = mode == "stream"
        self.bank = adist.ShardedFirBank(fir, C * world, device=local_rank,
                                         ring_slots=args.ring_slots if stream_mode else 0, fft_mult=args.fft_mult,
                                         sample_format=args.io, optimize_for="stream" if stream_mode else "batch")
        self.eng = eng = self.bank.engine
        assert eng.channels == C
        if args.effect != "none":
            from pyaudiodsptools_amd import config, effects
            config.initialize(args.fs, N)  # the tremolo reads its sampling rate from the package config
            eng.set_epilogue({"softclip": effects.CreateSoftClipper, "harddist": effects.CreateHardDistortion,
                              "saturator": effects.CreateSaturator, "volume": lambda: effects.CreateVolumeChange(-3.0),
                              "tremolo": effects.CreateTremolo}[args.effect]())
        self.stream = torch.cuda.current_stream(dev)
        sptr = self.stream.cuda_stream
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234 + rank)
        amp = float(os.environ.get("ADSP_BENCH_AMPLITUDE", "1"))  # tuning only: 0 = all-zero data (DVFS check)
        s16 = args.io == "s16"
        dt = torch.int16 if s16 else torch.float32

        def synth(shape):
            if s16:  # uniform 16-bit PCM at -6 dBFS
                return torch.randint(-16384, 16384, shape, device=dev, dtype=torch.int16, generator=gen)
            return torch.empty(shape, device=dev, dtype=torch.float32).uniform_(-amp, amp, generator=gen)
        if stream_mode:
            # zero-copy streaming: the synthetic producer has filled every ring slot before the timed region
            # (apply_device copies each batch into the ring and advances it; setup only)
            scratch = torch.empty((C, N), device=dev, dtype=dt)
            for _ in range(eng.ring_slots):
                batch = synth((C, N))
                eng.apply_device(batch, scratch, 1, sptr)
                torch.cuda.synchronize(dev)
            self.outs = [torch.empty((C, N), device=dev, dtype=dt) for _ in range(4)]
            self.spl = 1

            def run(k_steps):
                for i in range(k_steps):
                    eng.apply_ring(self.outs[i % 4], sptr)
        else:
            self.spl = spl = args.steps_per_launch
            # distinct resident input batches, > 256 MiB in total so the Infinity Cache cannot hold them
            n_in = max(2, min(8, -(-(768 << 20) // (spl * C * N * 4))))
            self.ins = [synth((spl, C, N)) for _ in range(n_in)]
            self.outs = [torch.empty((spl, C, N), device=dev, dtype=dt) for _ in range(2)]

            def run(k_steps):
                full, rest = divmod(k_steps, spl)
                for i in range(full):
                    eng.apply_device(self.ins[i % n_in], self.outs[i % 2], spl, sptr)
                if rest:  # exactly k_steps: one shorter launch at the end
                    eng.apply_device(self.ins[full % n_in][:rest], self.outs[full % 2][:rest], rest, sptr)
        self.run = run

    def measure(self, steps, warm, barrier=None, prewarm_ms=0.0, time_kernels=True):
        torch, eng = self.torch, self.eng
        steps = max(1, steps)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:  # clock ramp: untimed, same workload
            self.run(4 * self.spl)
            torch.cuda.synchronize()
        self.run(warm)
        torch.cuda.synchronize()
        if barrier:
            barrier()
        torch.cuda.synchronize()
        eng.enable_kernel_timing(time_kernels)
        t0 = time.perf_counter()
        self.run(steps)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if barrier:
            barrier()
        torch.cuda.synchronize()
        kern_ms, launches = eng.kernel_time()
        eng.enable_kernel_timing(False)
        chk = self.outs[0].reshape(-1)[:: max(1, self.outs[0].numel() // 65536)].float()
        assert bool(torch.isfinite(chk).all()) and (float(chk.abs().max()) > 0 or os.environ.get("ADSP_BENCH_AMPLITUDE") == "0")
        return steps, warm, wall, kern_ms, launches


def main():
    args = parse()
    import torch
    from pyaudiodsptools_amd import dist as adist

    rank, local_rank, world = adist.env_world()
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (pyaudiodsptools_amd has no CPU path)")
    # test hooks (single-GPU boxes): ADSP_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0, ADSP_BENCH_BACKEND=gloo
    # replaces RCCL - lets the N > 1 control flow run where only one GPU exists.  Never set by the driver.
    if os.environ.get("ADSP_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("ADSP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    barrier = None
    if world > 1:
        adist.init_process_group(backend)
        import torch.distributed as tdist
        barrier = tdist.barrier

    fir = make_fir(args)
    C, N = args.channels, args.chunk
    alg_bytes = ALG_BYTES_PER_SAMPLE if args.io == "f32" else 4
    main_run = Runner(args, args.mode, fir, dev, local_rank, world, rank)
    steps, warm, wall, kern_ms, launches = main_run.measure(args.steps, args.warmup, barrier, args.prewarm_ms)
    if world > 1:
        t = torch.tensor([wall, kern_ms], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        wall, kern_ms = float(t[0]), float(t[1])

    extra_stream = None
    if args.mode != "stream" and not args.no_stream_extra and world == 1:
        del main_run.ins
        torch.cuda.empty_cache()
        s_run = Runner(args, "stream", fir, dev, local_rank, world, rank)
        # wall clock without the per-launch timing events (two event records per 50 us launch are visible there), then a
        # shorter pass with them for the kernel duration
        s_steps, _, s_wall, _, _ = s_run.measure(2048, 512, None, args.prewarm_ms, time_kernels=False)
        _, _, _, s_kern_ms, s_launches = s_run.measure(512, 0, None, 0.0)
        s_per = s_kern_ms / 1e3 / s_launches
        extra_stream = {"value": round(C * N * s_steps / s_wall / 1e6, 1), "unit": "Msamples/s", "steps": s_steps,
                        "avg_kernel_us": round(s_per * 1e6, 2),
                        "roofline_frac": round(alg_bytes * C * N / s_per / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "one launch per step through the zero-copy ring (adsp_apply_ring), N outputs kept per 2N transform"}

    if rank == 0:
        eng = main_run.eng
        value = C * N * world * steps / wall / 1e6
        per_launch_s = kern_ms / 1e3 / launches
        samples_per_launch = C * N * steps / launches  # average: a --steps that is no multiple of the launch size ends short
        achieved = alg_bytes * samples_per_launch / per_launch_s / 1e9
        mode_key = "stream" if args.mode == "stream" else "batch"
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf)).get(f"{args.filter}_{C}x{N}_{mode_key}" + ("" if args.io == "f32" else "_s16"))
                if rec and rec.get("steps_per_launch", main_run.spl) == main_run.spl and steps % main_run.spl == 0:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": f"Msamples/s ({'float32' if args.io == 'f32' else 'int16 PCM'}, {N}-pt OLA FFT filter)",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warm,
            "ms_per_step": round(wall * 1e3 / steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.io == "f32" else "f32 arithmetic on s16 samples",
            "data": ("synthetic uniform(-1,1) float32" if args.io == "f32" else "synthetic uniform int16 PCM (-6 dBFS)") + " resident in HBM " +
                    ("(library input ring)" if args.mode == "stream" else "([steps, channels, chunk] batches)"),
            "config": {"workload": f"{FILTER_NAMES[args.filter]}{'' if args.effect == 'none' else ' -> ' + args.effect} @ {args.fs} Hz, {C} mono channels x {N}-sample chunks per GPU",
                       "channels_per_gpu": C, "chunk_size": N, "mode": mode_key, "steps_per_launch": main_run.spl,
                       "clock_ramp_prewarm_ms": args.prewarm_ms,
                       "fft_size": eng.geometry.fft_size, "spectrum": "real (zero-phase kernel)" if eng.real_spectrum else "complex",
                       "outputs_per_transform": (N if args.mode == "stream" else eng.block_outputs),
                       "parallelism": f"channel-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "adsp::fftconv_kernel", "avg_launch_us": round(per_launch_s * 1e6, 2),
                         "launches": launches, "algorithmic_bytes_per_launch": int(alg_bytes * samples_per_launch)},
        }
        if extra_stream:
            line["stream"] = extra_stream
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
