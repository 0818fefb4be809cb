#!/usr/bin/env python3
"""bench.py - throughput of the batched FFT filter hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): CreateLowCutFilter(800) @ 44.1 kHz on 4096 mono channels x
4096-sample chunks per GPU, float32, synthetic uniform(-1,1) input already resident in HBM.  A
"step" filters one [channels, chunk] batch (one chunk per channel, the reference's one apply() per
device), through the zero-copy streaming entry point (adsp_ring_acquire / adsp_apply_ring): one
kernel launch per step.  For N > 1 (torchrun, one rank per GPU) every rank owns its own channel
shard (weak scaling); the only collective is the RCCL broadcast of the filter spectrum before the
timed region.

Prints ONE JSON line on rank 0 (see the driver contract) including
  roofline     - algorithmic bytes (8 B/sample) / average kernel launch duration vs 8 TB/s
  cpu_baseline - the oracle's literal restatement of the reference (numpy, 1 core) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
ALG_BYTES_PER_SAMPLE = 8  # 4 B read + 4 B written, SURVEY.md 8(d)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--channels", type=int, default=4096, help="channels PER GPU")
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--fs", type=int, default=44100)
    ap.add_argument("--filter", default="lowcut", choices=["lowcut", "highcut", "eq3", "chain"])
    ap.add_argument("--mode", default="stream", choices=["stream", "offline"],
                    help="stream: one launch per step (zero-copy ring). offline: --steps-per-launch steps per launch")
    ap.add_argument("--steps-per-launch", type=int, default=16)
    ap.add_argument("--ring-slots", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


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
        a, b, c = orc.OracleLowCut(800, fs, n), orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n), orc.OracleHighCut(8000, fs, n)

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        adist.init_process_group("nccl")
        import torch.distributed as tdist

    fir = make_fir(args)
    C, N = args.channels, args.chunk
    offline = args.mode == "offline"
    bank = adist.ShardedFirBank(fir, C * world, device=local_rank, ring_slots=0 if offline else args.ring_slots)
    eng = bank.engine
    assert eng.channels == C
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    n_out_bufs = 4
    if offline:
        spl = args.steps_per_launch
        # resident input: enough distinct launches' worth of chunks to defeat the 256 MiB Infinity Cache
        n_in = max(2, min(8, (1 << 30) // (spl * C * N * 4)))
        ins = [torch.empty((spl, C, N), device=dev, dtype=torch.float32).uniform_(-1, 1, generator=gen) for _ in range(n_in)]
        outs = [torch.empty((spl, C, N), device=dev, dtype=torch.float32) for _ in range(2)]
        steps = (args.steps // spl) * spl or spl
        warm = -(-args.warmup // spl) * spl if args.warmup else 0

        def run(k_steps):
            for i in range(k_steps // spl):
                eng.apply_device(ins[i % n_in], outs[i % 2], spl, sptr)
        launches = steps // spl
    else:
        # zero-copy streaming: the synthetic producer has already filled every ring slot
        # (apply_device copies each new batch into the ring and advances it; setup, untimed)
        scratch = torch.empty((C, N), device=dev, dtype=torch.float32)
        for _ in range(eng.ring_slots):
            batch = torch.empty((C, N), device=dev, dtype=torch.float32).uniform_(-1, 1, generator=gen)
            eng.apply_device(batch, scratch, 1, sptr)
            torch.cuda.synchronize(dev)
        outs = [torch.empty((C, N), device=dev, dtype=torch.float32) for _ in range(n_out_bufs)]
        steps, warm = args.steps, args.warmup

        def run(k_steps):
            for i in range(k_steps):
                eng.apply_ring(outs[i % n_out_bufs], sptr)
        launches = steps

    run(warm)
    torch.cuda.synchronize(dev)
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    run(steps)
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize(dev)
    gpu_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([wall, gpu_ms], device=dev, dtype=torch.float64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        wall, gpu_ms = float(t[0]), float(t[1])

    # sanity: outputs are finite and non-trivial
    chk = outs[0].reshape(-1)[:: max(1, outs[0].numel() // 65536)]
    assert bool(torch.isfinite(chk).all()) and float(chk.abs().max()) > 0

    if rank == 0:
        samples_per_step = C * N * world
        value = samples_per_step * steps / wall / 1e6
        per_launch_s = gpu_ms / 1e3 / launches
        samples_per_launch = C * N * (steps // launches)
        achieved = ALG_BYTES_PER_SAMPLE * samples_per_launch / per_launch_s / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf)).get(f"{args.filter}_{C}x{N}_{args.mode}")
                if rec:
                    traffic = rec
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/s (float32, 4096-pt OLA FFT filter)" if N == 4096 else f"Msamples/s (float32, {N}-pt OLA FFT filter)",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warm,
            "ms_per_step": round(wall * 1e3 / steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic uniform(-1,1) float32, resident in HBM (library input ring)" if not offline else
                    "synthetic uniform(-1,1) float32, resident in HBM ([steps, C, N] batches)",
            "config": {"workload": f"Create{ {'lowcut':'LowCutFilter(800)','highcut':'HighCutFilter(8000)','eq3':'EQ3BandFFT(100,2,700,-4,8000,5)','chain':'LowCut(800)->EQ3BandFFT->HighCut(8000) fused'}[args.filter]} "
                                   f"@ {args.fs} Hz, {C} mono channels x {N}-sample chunks per GPU",
                       "channels_per_gpu": C, "chunk_size": N, "mode": args.mode,
                       "steps_per_launch": (args.steps_per_launch if offline else 1),
                       "fft_size": eng.geometry.fft_size, "outputs_per_transform": (eng.block_outputs if offline else N),
                       "parallelism": f"channel-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "fftconv_kernel", "avg_launch_us": round(per_launch_s * 1e6, 2),
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_SAMPLE * samples_per_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
