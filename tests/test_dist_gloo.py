"""Multi-process (gloo, world_size 2, CPU) tests of the N-GPU layer: contiguous channel shards, ONE
broadcast of the filter spectrum from rank 0, zero steady-state communication.  The HIP engine is
replaced by a numpy overlap-save stand-in that consumes the BROADCAST spectrum, so a wrong shard
range or a spectrum that did not arrive shows up as a parity failure against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_parity


def test_shard_range_partitions_exactly():
    from pyaudiodsptools_amd.dist import shard_range
    for total in (1, 2, 7, 8, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(65536, 8, 3) == (3 * 8192, 4 * 8192)  # config 4: 8192 channels per GPU


class NumpyEngine:
    """Stand-in for FirEngine on machines without a GPU: same geometry, spectrum supplied from outside."""

    def __init__(self, fir, channels=1, device=0, ring_slots=0, fft_mult=0):
        from pyaudiodsptools_amd.design import overlap_save_geometry
        self.geometry = overlap_save_geometry(fir, fft_mult)
        self.channels, self.n = channels, fir.chunk_size
        self.hist = np.zeros((channels, self.geometry.history_chunks * self.n))
        self.spec = None

    def upload_spectrum(self, spectrum_f32, reach=None):
        self.spec = np.asarray(spectrum_f32, np.float32).view(np.complex64).astype(np.complex128)

    def apply_host(self, x):  # x [C, N]
        g, n = self.geometry, self.n
        buf = np.concatenate([self.hist, x.astype(np.float64), np.zeros((self.channels, g.fft_size))], axis=1)
        a = self.hist.shape[1] - g.lookback
        y = np.fft.irfft(np.fft.rfft(buf[:, a:a + g.fft_size], axis=1) * self.spec, g.fft_size, axis=1)
        self.hist = np.concatenate([self.hist, x], axis=1)[:, -self.hist.shape[1]:]
        return y[:, g.out_offset:g.out_offset + n].astype(np.float32)


def _worker(rank, world, port, total_channels, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from pyaudiodsptools_amd import design
    from pyaudiodsptools_amd.dist import ShardedFirBank, init_process_group
    init_process_group("gloo")
    n, fs, steps = 512, 44100, 5
    # only rank 0 holds the real design; other ranks start from a WRONG filter to prove the broadcast is used
    cutoff = 800 if rank == 0 else 5000
    fir = design.FirStream(design.lowcut_kernel(cutoff, fs, n), n)
    bank = ShardedFirBank(fir, total_channels, device=0, engine_factory=NumpyEngine)
    x = np.random.default_rng(99).uniform(-1, 1, (steps, total_channels, n)).astype(np.float32)
    mine = x[:, bank.lo:bank.hi]
    if bank.engine is None:  # more ranks than channels: an idle rank that still took part in the broadcasts
        y = np.zeros((steps, 0, n), np.float32)
    else:
        y = np.stack([bank.engine.apply_host(mine[k]) for k in range(steps)])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=bank.lo, hi=bank.hi, y=y, spec=bank.spectrum)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total_channels", [1, 6, 7])
def test_two_rank_sharded_filter_matches_oracle(tmp_path, total_channels):
    import torch.multiprocessing as mp
    from oracle import fftfilter_oracle as orc
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total_channels, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert int(parts[0]["lo"]) == 0 and int(parts[-1]["hi"]) == total_channels and int(parts[0]["hi"]) == int(parts[1]["lo"])
    assert np.array_equal(parts[0]["spec"], parts[1]["spec"])  # bit-identical spectra on every rank
    y = np.concatenate([p["y"] for p in parts], axis=1)
    n, steps = 512, 5
    x = np.random.default_rng(99).uniform(-1, 1, (steps, total_channels, n)).astype(np.float32)
    for c in range(total_channels):
        ref = orc.OracleLowCut(800, 44100, n)
        want = np.concatenate([ref.apply(x[k, c]) for k in range(steps)])
        assert_parity(y[:, c].reshape(-1), want, what=f"channel {c}")


def test_local_bank_shards_channels_over_devices_in_one_process():
    """dist.LocalFirBank (one process, N devices, adsp_bcast_spectrum) with stand-in engines: contiguous balanced shards, the
    root's filter on every device (device 1 starts from a WRONG filter), more devices than channels."""
    from pyaudiodsptools_amd import design
    from pyaudiodsptools_amd.dist import LocalFirBank
    n, fs, steps = 512, 44100, 4
    fir = design.FirStream(design.lowcut_kernel(800, fs, n), n)
    wrong = design.FirStream(design.lowcut_kernel(5000, fs, n), n)
    made = []

    class Chunked(NumpyEngine):  # [steps, C, N] batches like FirEngine.apply_host
        def apply_host(self, x):
            return np.stack([NumpyEngine.apply_host(self, x[k]) for k in range(x.shape[0])])

    def factory(f, channels=1, device=0, ring_slots=0, **kw):
        mine = wrong if device == 1 else f
        e = Chunked(mine, channels=channels, device=device)
        e.upload_spectrum(design.engine_spectrum(mine, design.overlap_save_geometry(mine)))
        made.append(e)
        return e

    def broadcast(engines, root):
        for e in engines:
            e.spec = engines[root].spec.copy()

    for total, devices in ((7, [0, 1, 2]), (2, [0, 1, 2, 3])):
        made.clear()
        bank = LocalFirBank(fir, total, devices=devices, engine_factory=factory, broadcast=broadcast)
        assert [hi - lo for lo, hi in bank.shards] == ([3, 2, 2] if total == 7 else [1, 1, 0, 0])
        assert sum(e is not None for e in bank.engines) == min(total, len(devices))
        x = np.random.default_rng(total).uniform(-1, 1, (steps, total, n)).astype(np.float32)
        y = bank.apply_host(x)
        from oracle import fftfilter_oracle as o
        taps = o.lowcut_taps(800, fs, n)
        for c in range(total):
            ref = o.direct_stream_convolution(taps, x[:, c].reshape(-1), n)
            assert np.abs(y[:, c].reshape(-1) - ref).max() <= 1e-5 * np.abs(ref).max()
        bank.close()


# ---- round 4: the collective behind the C ABI (adsp_bcast_spectrum_rank) - host logic only, no GPU ---------------------
def _uid_worker(rank, world, path, q):
    from pyaudiodsptools_amd import dist
    uid = dist.exchange_unique_id(rank, world, lambda: bytes(range(128)), path=path, timeout=60.0)
    q.put((rank, uid))
    if rank == 0:  # rank 0 removes the file when it exits; in a real job the collective that follows keeps it alive until
        import time  # every rank has joined - here it simply lingers
        time.sleep(5.0)


def test_unique_id_exchange_through_a_file_two_processes(tmp_path):
    """exchange_unique_id: rank 0 draws the 128 bytes and publishes them atomically, rank 1 (started FIRST, so it has to
    wait) picks them up; one exchange per process."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "rccl_id")
    p1 = ctx.Process(target=_uid_worker, args=(1, 2, path, q))
    p1.start()
    import time
    time.sleep(0.3)
    p0 = ctx.Process(target=_uid_worker, args=(0, 2, path, q))
    p0.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    p0.join(30)
    p1.join(30)
    assert got[0] == got[1] == bytes(range(128))


class _FakeAbiEngine:
    """Stand-in for FirEngine on a box without a GPU: records what the bank does with it."""
    calls = []

    def __init__(self, fir, channels=1, device=0, ring_slots=0, **kw):
        self.fir, self.channels, self.kw = fir, channels, kw
        self.spectrum = None

    def bcast_rank(self, uid, rank, world, root=0):
        _FakeAbiEngine.calls.append((bytes(uid), rank, world, root, self.channels))
        self.spectrum = np.arange(4, dtype=np.float32)

    def close(self):
        self.closed = True


def test_sharded_bank_abi_carrier_host_logic(monkeypatch):
    """carrier="abi": the bank builds its engine first, fetches the job's id once and calls bcast_rank; no
    torch.distributed involved."""
    from pyaudiodsptools_amd import FirStream, design, dist
    import pyaudiodsptools_amd.engine as engine_mod
    monkeypatch.setattr(dist, "_job_unique_ids", {})
    monkeypatch.setattr(engine_mod, "rccl_unique_id", lambda: b"\x07" * 128)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    _FakeAbiEngine.calls.clear()
    fir = FirStream(design.lowcut_kernel(300, 44100, 512), 512)
    bank = dist.ShardedFirBank(fir, 10, device=0, engine_factory=_FakeAbiEngine, carrier="abi", optimize_for="batch")
    assert bank.carrier == "abi" and bank.engine.channels == 10 and bank.engine.kw == {"optimize_for": "batch"}
    assert _FakeAbiEngine.calls == [(b"\x07" * 128, 0, 1, 0, 10)]
    assert np.array_equal(bank.spectrum, np.arange(4, dtype=np.float32))
    with pytest.raises(ValueError):
        dist.ShardedFirBank(fir, 10, engine_factory=_FakeAbiEngine, carrier="mpi")


def test_unique_id_file_exchange_ignores_leftovers_and_is_keyed_by_path(tmp_path, monkeypatch):
    """the id file carries a magic word, must belong to this user and must be fresh; rank 0 replaces whatever an earlier
    (killed) attempt left under the name; the per-process cache is keyed by the file, and every elastic attempt has its own name."""
    import os
    import threading
    import time as _time
    from pyaudiodsptools_amd import dist
    monkeypatch.setattr(dist, "_job_unique_ids", {})
    path = str(tmp_path / "id")
    open(path, "wb").write(b"\x01" * 200)  # a leftover without the magic word: never accepted
    with pytest.raises(TimeoutError):
        dist.exchange_unique_id(1, 2, None, path=path, timeout=0.3)
    stale = str(tmp_path / "stale")
    open(stale, "wb").write(dist._ID_MAGIC + b"0" * 30 + b"\x02" * 128)
    os.utime(stale, (_time.time() - 4000, _time.time() - 4000))  # well formed, but from a job of long ago
    with pytest.raises(TimeoutError):
        dist.exchange_unique_id(1, 2, None, path=stale, timeout=0.3)
    got = {}
    reader = threading.Thread(target=lambda: got.setdefault("uid", dist.exchange_unique_id(1, 2, None, path=path, timeout=20.0)))
    reader.start()
    _time.sleep(0.2)
    monkeypatch.setattr(dist, "_job_unique_ids", {})  # (rank 0 is another process in real life: its own cache)
    uid0 = dist.exchange_unique_id(0, 2, lambda: b"\x07" * 128, path=path)
    reader.join(20.0)
    assert uid0 == b"\x07" * 128 and got["uid"] == uid0
    assert (os.stat(path).st_mode & 0o777) == 0o600
    assert dist.exchange_unique_id(0, 2, lambda: b"\x09" * 128, path=path) == uid0          # cached per path ...
    assert dist.exchange_unique_id(0, 1, lambda: b"\x09" * 128, path=path + "b") == b"\x09" * 128  # ... another path, another id
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    a = dist._id_file_path()
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    assert dist._id_file_path() != a and "job7" in a


def test_local_bank_float64_engines_keep_their_filter_through_the_broadcast():
    """sample_format='s16_f64' with the banks.  LocalFirBank hands the engines to broadcast_filter untouched (the float64
    spectrum travels as float64 inside adsp_bcast_spectrum: no float32 re-upload), ShardedFirBank checks cross-rank agreement on
    the float32 image and leaves the float64 tables alone."""
    from pyaudiodsptools_amd import FirStream, design, dist
    seen = []

    class Eng:
        def __init__(self, fir, channels=1, device=0, ring_slots=0, **kw):
            self.kw, self.channels, self.uploads = kw, channels, 0

        def upload_spectrum(self, *a, **k):
            self.uploads += 1

        def upload_spectrum_device(self, *a, **k):
            self.uploads += 1

    fir = FirStream(design.lowcut_kernel(300, 44100, 512), 512)
    bank = dist.LocalFirBank(fir, 10, devices=[0, 1], sample_format="s16_f64", engine_factory=Eng, broadcast=lambda engines, root: seen.append((len(engines), root)))
    assert seen == [(2, 0)] and all(e.kw == {"sample_format": "s16_f64"} and e.uploads == 0 for e in bank.engines)
    bank.close()
    sb = dist.ShardedFirBank(fir, 4, engine_factory=Eng, sample_format="s16_f64", carrier="torch")
    assert sb.engine.uploads == 0 and sb.engine.kw == {"sample_format": "s16_f64"}


# ---- kernels longer than one transform shard too (round 6): ShardedFirBank -> the uniformly partitioned engine, its partition spectra
# are what the one broadcast carries -----------------------------------------------------------------------------------------------
class NumpyUpolsEngine:
    """Stand-in for UpolsFirEngine: tests/test_upols_host.py's numpy mirror of the engine's block bookkeeping per channel, the partition
    spectra supplied from outside (set_spectra: what the broadcast delivered)."""

    def __init__(self, fir, channels=1, device=0, block=8192, partition=None, **kw):
        from test_upols_host import UpolsMirror
        self.mirrors = [UpolsMirror(fir, block, 1) for _ in range(channels)]
        self.block, self.n = block, fir.chunk_size

    def set_spectra(self, spectra_f32):
        sp = np.asarray(spectra_f32, np.float32).reshape(-1, self.block + 1, 2)
        for m in self.mirrors:
            assert sp.shape[0] == m.P
            m.H = sp[..., 0].astype(np.float64) + 1j * sp[..., 1].astype(np.float64)

    def apply_host(self, x):  # [C, N]
        return np.stack([m.apply(x[c]) for c, m in enumerate(self.mirrors)]).astype(np.float32)


def _long_worker(rank, world, port, total_channels, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from pyaudiodsptools_amd import design
    from pyaudiodsptools_amd.dist import ShardedFirBank, init_process_group
    init_process_group("gloo")
    n, steps, taps_len = 20000, 3, 40001
    rng = np.random.default_rng(5 if rank == 0 else 6)   # only rank 0 holds the real kernel
    taps = rng.standard_normal(taps_len) * np.hanning(taps_len) / taps_len ** 0.5
    fir = design.FirStream(taps, n, latency_chunks=2, lookahead=20000)
    bank = ShardedFirBank(fir, total_channels, device=0, engine_factory=NumpyUpolsEngine)
    x = np.random.default_rng(98).uniform(-1, 1, (steps, total_channels, n)).astype(np.float32)
    mine = x[:, bank.lo:bank.hi]
    y = np.zeros((steps, 0, n), np.float32) if bank.engine is None else np.stack([bank.engine.apply_host(mine[k]) for k in range(steps)])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=bank.lo, hi=bank.hi, y=y, spec=bank.spectrum)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_channels", [1, 3])
def test_two_rank_sharded_long_kernel_matches_the_direct_convolution(tmp_path, total_channels):
    import torch.multiprocessing as mp
    from oracle import fftfilter_oracle as orc
    world = 2
    mp.spawn(_long_worker, args=(world, _free_port(), total_channels, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert int(parts[0]["lo"]) == 0 and int(parts[-1]["hi"]) == total_channels and int(parts[0]["hi"]) == int(parts[1]["lo"])
    assert np.array_equal(parts[0]["spec"], parts[1]["spec"]) and parts[0]["spec"].size == 5 * 2 * 8193  # 40001 taps: 5 partitions of 8192
    y = np.concatenate([p["y"] for p in parts], axis=1)
    n, steps, taps_len = 20000, 3, 40001
    taps = np.random.default_rng(5).standard_normal(taps_len) * np.hanning(taps_len) / taps_len ** 0.5
    x = np.random.default_rng(98).uniform(-1, 1, (steps, total_channels, n)).astype(np.float32)
    for c in range(total_channels):
        want = orc.direct_stream_convolution(taps, x[:, c].reshape(-1), n, 2, 20000)
        assert_parity(y[:, c].reshape(-1), want, what=f"channel {c}")


def test_finalize_forgets_the_unique_ids(tmp_path, monkeypatch):
    """rccl_finalize destroys the communicators, so the ids cached per id file must go with them."""
    from pyaudiodsptools_amd import dist
    path = str(tmp_path / "id")
    calls = []

    def make_id():
        calls.append(1)
        return bytes([len(calls)]) * 128
    a = dist.exchange_unique_id(0, 1, make_id, path=path)
    assert dist.exchange_unique_id(0, 1, make_id, path=path) == a and len(calls) == 1   # cached
    dist.forget_unique_ids()
    b = dist.exchange_unique_id(0, 1, make_id, path=path)
    assert b != a and len(calls) == 2
    dist.forget_unique_ids()
