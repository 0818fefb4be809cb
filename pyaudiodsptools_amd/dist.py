"""Multi-GPU layer: channels shard, nothing else moves.

Channels never interact (each reference device instance is private state, Example2.py:13-21), so
the N-GPU job is N independent engines over contiguous channel ranges.  The only collective is ONE
broadcast of the filter spectrum (complex64, 32 KiB at N = 4096) from rank 0 at filter
creation/change.  One process per GPU, launched by torchrun / torch.distributed.run.  Two carriers:

  "torch"  torch.distributed (backend "nccl" IS RCCL on ROCm; gloo in the CPU tests), the spectrum then goes into the
           engine with adsp_set_spectrum_device - the default when a process group exists;
  "abi"    adsp_bcast_spectrum_rank: RCCL opened by libadsp itself (ncclCommInitRank), the 128-byte id handed from
           rank 0 to the others through a small file - no torch.distributed anywhere (the default when WORLD_SIZE > 1
           and no process group was initialised; force with ADSP_DIST_CARRIER=abi).
"""
import os
import time

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not distributed."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_channels, world_size, rank):
    """Contiguous, balanced channel range [lo, hi) owned by `rank` (first ranks take the remainder)."""
    base, rem = divmod(int(n_channels), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend=None):
    """Initialise torch.distributed from the environment (127.0.0.1 rendezvous by default)."""
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")        # a single process outside torchrun is a world of one
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend is None:
        import torch
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend)
    return dist


_job_unique_ids = {}  # id file path -> the 128 bytes this process took part with (one communicator per id, cached by libadsp)
_ID_MAGIC = b"ADSPRCCL1"


def forget_unique_ids():
    """Drop the cached ids (engine.rccl_finalize calls this: the communicators built from them no longer exist) and remove the id
    files this process wrote as rank 0."""
    for path in list(_job_unique_ids):
        _job_unique_ids.pop(path, None)
        if path in _files_written:
            _files_written.discard(path)
            try:
                os.remove(path)
            except OSError:
                pass


_files_written = set()


def _id_file_path(path=None):
    """`path`, or $ADSP_RCCL_ID_FILE, or <tmp>/adsp_rccl_<MASTER_PORT>_<parent pid>_<run id>_<restart count>: ranks started by one
    torchrun agent share the parent pid and the port; the elastic run id and restart count (TORCHELASTIC_RUN_ID /
    TORCHELASTIC_RESTART_COUNT, set by torchrun) make every ATTEMPT of a job use a name of its own, so a restarted worker group never
    polls the file of the attempt that died."""
    import tempfile
    if path or os.environ.get("ADSP_RCCL_ID_FILE"):
        return path or os.environ["ADSP_RCCL_ID_FILE"]
    run = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum() or ch in "-_")[:40]
    return os.path.join(tempfile.gettempdir(), f"adsp_rccl_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{run}_"
                                               f"{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}")


def exchange_unique_id(rank, world, make_id, path=None, timeout=120.0, max_age=300.0):
    """Hand the 128 bytes `make_id()` returns on rank 0 to every rank of a one-process-per-GPU job - once per process and id file
    (the communicator behind it is cached by libadsp and reused for every later broadcast).

    The carrier is a small file (_id_file_path) - NODE-LOCAL by default: a multi-node job points ADSP_RCCL_ID_FILE at shared storage
    (or hands the id over any other way and calls FirEngine.bcast_rank itself).  Rank 0 removes whatever is left under that name,
    then creates the file with O_EXCL and mode 0600 under a temporary name and renames it into place; it holds a magic word, rank
    0's pid and start time, and the id.  The other ranks accept a file only if it is owned by their own user, carries the magic
    word and is not older than `max_age` seconds (a file left behind by a killed job of long ago is ignored, not joined)."""
    path = _id_file_path(path)
    if path in _job_unique_ids:
        return _job_unique_ids[path]
    if world == 1:
        _job_unique_ids[path] = bytes(make_id())
        return _job_unique_ids[path]
    if rank == 0:
        uid = bytes(make_id())
        try:
            os.remove(path)  # a leftover of an attempt that was killed (atexit did not run): nobody may read it as this attempt's
        except FileNotFoundError:
            pass
        tmp = f"{path}.{os.getpid()}.tmp"
        try:
            os.remove(tmp)
        except FileNotFoundError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as fh:
            fh.write(_ID_MAGIC + f"{os.getpid():010d}{time.time():020.3f}".encode() + uid)
        os.replace(tmp, path)
        _files_written.add(path)
        import atexit
        atexit.register(lambda: os.path.exists(path) and os.remove(path))
    else:
        t0 = time.time()
        head = len(_ID_MAGIC) + 30
        while True:
            try:
                st = os.stat(path)
                with open(path, "rb") as fh:
                    blob = fh.read()
                fresh = time.time() - st.st_mtime <= max_age
                if st.st_uid == os.getuid() and fresh and blob.startswith(_ID_MAGIC) and len(blob) >= head + 128:
                    uid = blob[head:head + 128]
                    break
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout:
                raise TimeoutError(f"rank {rank}: no RCCL id from rank 0 in {path} after {timeout:.0f} s (the file is node-local: a "
                                   "multi-node job sets ADSP_RCCL_ID_FILE to a path on shared storage)")
            time.sleep(0.01)
    _job_unique_ids[path] = uid
    return uid


def broadcast_spectrum(spectrum_f32, src=0, device=None):
    """Broadcast the interleaved float32 spectrum from `src`; returns (tensor, numpy copy).

    Non-source ranks pass an array of the right size (contents ignored) or just its length.
    With backend nccl the tensor lives on `device` (RCCL broadcast over xGMI)."""
    import torch
    import torch.distributed as dist
    if isinstance(spectrum_f32, int):
        spectrum_f32 = np.zeros(spectrum_f32, np.float32)
    t = torch.from_numpy(np.ascontiguousarray(spectrum_f32, dtype=np.float32).copy())
    if device is not None:
        t = t.to(device)
    if dist.is_initialized():  # also at world size 1: the collective path is the same code on one GPU and on eight
        dist.broadcast(t, src=src)
    return t, t.detach().cpu().numpy()


class ShardedFirBank:
    """All channels of a job, this rank's shard on this rank's GPU.

    rank 0 designs the filter; every rank receives the identical spectrum by broadcast and uploads
    it to its own engine (bit-identical spectra across ranks by construction)."""

    def __init__(self, fir, total_channels, device=0, ring_slots=0, engine_factory=None, fft_mult=0,
                 sample_format="f32", optimize_for="stream", carrier=None):
        from .design import engine_spectrum, overlap_save_geometry
        rank, _, world = env_world()
        pg = False
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                rank, world, pg = dist.get_rank(), dist.get_world_size(), True
        except ImportError:
            pass
        self.rank, self.world = rank, world
        self.lo, self.hi = shard_range(total_channels, world, rank)
        self.total_channels = int(total_channels)
        carrier = carrier or os.environ.get("ADSP_DIST_CARRIER") or ("torch" if pg or world == 1 else "abi")
        if carrier not in ("torch", "abi"):
            raise ValueError("carrier must be 'torch' or 'abi'")
        self.carrier = carrier
        from .design import fits_one_transform
        if not fits_one_transform(fir):
            # a kernel longer than one transform (the reference's own GPU example: chunk 88200, Example4.py:5 / ModuleTestsGPU.py:35): the
            # uniformly partitioned engine, whose filter is its partition spectra - the same ONE broadcast, of [partitions][block + 1] bins
            self._init_long(fir, device, sample_format, engine_factory, pg)
            return
        geo = overlap_save_geometry(fir, fft_mult, optimize_for)
        kw = dict(**({'fft_mult': fft_mult} if fft_mult else {}), **({'sample_format': sample_format} if sample_format != "f32" else {}),
                  **({'optimize_for': optimize_for} if optimize_for != "stream" else {}))
        if carrier == "abi":
            # the collective lives behind the C ABI: every rank builds its engine (a rank without channels a one-channel stand-in,
            # it still has to take part), rank 0's filter reaches the others through adsp_bcast_spectrum_rank
            from .engine import FirEngine, rccl_unique_id
            factory = engine_factory or FirEngine
            eng = factory(fir, channels=max(1, self.hi - self.lo), device=device, ring_slots=ring_slots, **kw)
            uid = exchange_unique_id(rank, world, rccl_unique_id)
            eng.bcast_rank(uid, rank, world, 0)
            self.spectrum = eng.spectrum
            self.spectrum_tensor = None
            if self.hi == self.lo:
                if hasattr(eng, "close"):
                    eng.close()
                eng = None
            self.engine = eng
            return
        n_floats = 2 * (geo.fft_size // 2 + 1)
        from .design import PCM16_GAIN
        gain = PCM16_GAIN if sample_format == "s16" else 1.0
        spec = engine_spectrum(fir, geo, gain) if rank == 0 else np.zeros(n_floats, np.float32)
        bdev = None
        try:
            import torch
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_backend() == "nccl":
                bdev = torch.device("cuda", device)
        except ImportError:
            pass
        self.spectrum_tensor, self.spectrum = broadcast_spectrum(spec, 0, bdev)
        # the spectrum only means something with rank 0's window geometry: every rank must have derived the same one
        mine = np.array([geo.fft_size, geo.history_chunks, geo.lookback, geo.out_offset, geo.shift, geo.max_block_outputs,
                         int(geo.zero_phase)], dtype=np.float32)
        _, theirs = broadcast_spectrum(mine, 0, bdev)
        if not np.array_equal(mine, theirs):
            raise ValueError(f"rank {rank}: filter geometry {mine.astype(int).tolist()} differs from rank 0's "
                             f"{theirs.astype(int).tolist()} - every rank must be constructed with the same kind of filter")
        if engine_factory is None:
            from .engine import FirEngine
            engine_factory = FirEngine
        if self.hi == self.lo:  # more ranks than channels: this rank idles (it still took part in the broadcasts)
            self.engine = None
            return
        self.engine = engine_factory(fir, channels=self.hi - self.lo, device=device, ring_slots=ring_slots, **kw)
        if sample_format == "s16_f64":
            # the exact-FFT engines keep float64 tables: the float32 broadcast above is the cross-rank agreement check only
            # (every rank designs the same filter in float64; replacing the tables with the float32 copy would throw the
            # engine's precision away while it still ran at a third of the rate)
            mine32 = engine_spectrum(fir, geo, gain)
            if not np.array_equal(mine32, self.spectrum):
                raise ValueError(f"rank {rank}: this rank's filter differs from rank 0's - float64 engines need the same FirStream on every rank")
        elif bdev is not None and hasattr(self.engine, "upload_spectrum_device"):
            # RCCL path: the spectrum the collective left in this GPU's memory goes straight into the engine
            # (adsp_set_spectrum_device); the collective ran on torch's current stream, which is drained first
            import torch
            torch.cuda.current_stream(bdev).synchronize()
            self.engine.upload_spectrum_device(self.spectrum_tensor, n_floats // 2, reach=max(0, -geo.shift))
        else:
            # gloo / no process group: `self.spectrum` is the host copy taken after the collective completed
            self.engine.upload_spectrum(self.spectrum, reach=max(0, -geo.shift))


def _sharded_init_long(self, fir, device, sample_format, engine_factory, pg):
    """ShardedFirBank for kernels longer than one transform: every rank builds its UpolsFirEngine (a rank without channels a one-channel
    stand-in under the "abi" carrier - it still has to take part), rank 0's partition spectra reach the others through the carrier,
    together with the tuple (chunk, block, partitions, delay) they belong to."""
    from .design import PCM16_GAIN, choose_uniform_block, partition_uniform
    from .engine import UpolsFirEngine, rccl_unique_id, upols_supported
    if sample_format not in ("f32", "s16"):
        raise ValueError("kernels longer than one transform: float32 or int16 samples")
    if engine_factory is None and not upols_supported(fir):
        raise ValueError("a sharded bank of kernels longer than one transform needs a chunk size that is a multiple of 4 and a delay of at least one block")
    factory = engine_factory or UpolsFirEngine
    # the block size is chosen for the JOB's channel count, so that every rank - whatever its shard - partitions alike
    block = choose_uniform_block(fir, self.total_channels, UpolsFirEngine.block_sizes() if engine_factory is None else (8192, 16384))
    kw = {"sample_format": sample_format} if sample_format != "f32" else {}
    if self.carrier == "abi":
        eng = factory(fir, channels=max(1, self.hi - self.lo), device=device, block=block, **kw)
        eng.bcast_rank(exchange_unique_id(self.rank, self.world, rccl_unique_id), self.rank, self.world, 0)
        self.spectrum, self.spectrum_tensor = eng.get_spectra(), None
        if self.hi == self.lo:
            eng.close()
            eng = None
        self.engine = eng
        return
    gain = PCM16_GAIN if sample_format == "s16" else 1.0
    part = partition_uniform(fir, block, gain)
    mine = np.array([fir.chunk_size, part.block, part.n_partitions, part.delay], dtype=np.float32)
    spec = np.ascontiguousarray(part.spectra, dtype=np.float32).reshape(-1) if self.rank == 0 else np.zeros(part.spectra.size, np.float32)
    bdev = None
    try:
        import torch
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_backend() == "nccl":
            bdev = torch.device("cuda", device)
    except ImportError:
        pass
    _, theirs = broadcast_spectrum(mine, 0, bdev)
    if not np.array_equal(mine, theirs):
        raise ValueError(f"rank {self.rank}: partitioning {mine.astype(int).tolist()} differs from rank 0's {theirs.astype(int).tolist()} - every rank "
                         "must be constructed with the same kind of filter")
    self.spectrum_tensor, self.spectrum = broadcast_spectrum(spec, 0, bdev)
    if self.hi == self.lo:
        self.engine = None
        return
    self.engine = factory(fir, channels=self.hi - self.lo, device=device, block=block, partition=part, **kw)
    self.engine.set_spectra(self.spectrum)  # what the collective delivered (rank 0: its own design), bit-identical on every rank


ShardedFirBank._init_long = _sharded_init_long


class LocalFirBank:
    """All channels of a job on the GPUs of ONE process: contiguous channel shards, one engine per device, one host thread per
    device, and the filter shared by `adsp_bcast_spectrum` - the RCCL broadcast inside libadsp (ncclCommInitAll: no torchrun,
    no torch.distributed).  The torchrun counterpart is ShardedFirBank.

    devices: list of HIP device ordinals (default: every visible GPU).  `engine_factory` / `broadcast` exist for the CPU
    tests (stand-ins for FirEngine / engine.broadcast_filter)."""

    def __init__(self, fir, total_channels, devices=None, ring_slots=0, fft_mult=0, sample_format="f32", optimize_for="stream",
                 engine_factory=None, broadcast=None):
        from concurrent.futures import ThreadPoolExecutor
        if devices is None:
            from . import _capi
            devices = list(range(max(1, _capi.device_count())))
        if engine_factory is None:
            from .design import choose_uniform_block, fits_one_transform
            from .engine import FirEngine, UpolsFirEngine, broadcast_filter, make_engine
            broadcast = broadcast or broadcast_filter
            if fits_one_transform(fir):
                engine_factory = FirEngine
            else:  # kernels longer than one transform shard too (Example4's chunk 88200): one block size for the whole job
                job_block = choose_uniform_block(fir, int(total_channels), UpolsFirEngine.block_sizes())

                def engine_factory(f, **kw):
                    return make_engine(f, block=job_block, **kw)
        self.devices = [int(d) for d in devices]
        self.total_channels = int(total_channels)
        self.shards = [shard_range(total_channels, len(self.devices), i) for i in range(len(self.devices))]
        kw = {}
        if fft_mult:
            kw["fft_mult"] = fft_mult
        if sample_format != "f32":
            kw["sample_format"] = sample_format
        if optimize_for != "stream":
            kw["optimize_for"] = optimize_for
        # engines of devices without channels (more GPUs than channels) are not created
        self.engines = [engine_factory(fir, channels=hi - lo, device=d, ring_slots=ring_slots, **kw) if hi > lo else None
                        for d, (lo, hi) in zip(self.devices, self.shards)]
        live = [e for e in self.engines if e is not None]
        if broadcast is not None and live:
            broadcast(live, 0)  # every engine takes over engine 0's filter (bit-identical tables on every GPU)
        self._pool = ThreadPoolExecutor(max_workers=max(1, len(live)))

    def apply_host(self, x):
        """x [steps, total_channels, N] (or [total_channels, N]) host array -> same shape; every shard on its own GPU, concurrently."""
        x = np.asarray(x)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.shape[1] != self.total_channels:
            raise ValueError(f"expected {self.total_channels} channels, got {x.shape[1]}")
        jobs = [(e, lo, hi) for e, (lo, hi) in zip(self.engines, self.shards) if e is not None]
        parts = list(self._pool.map(lambda j: j[0].apply_host(np.ascontiguousarray(x[:, j[1]:j[2]])), jobs))
        out = np.concatenate(parts, axis=1)
        return out[0] if squeeze else out

    def reset(self):
        for e in self.engines:
            if e is not None:
                e.reset()

    def close(self):
        self._pool.shutdown(wait=True)
        for e in self.engines:
            if e is not None and hasattr(e, "close"):
                e.close()
