"""CPU-baseline timing of the oracle (bench.py's `cpu_baseline` leg ONLY - test infrastructure, never the product path).

BASELINE.md section 4 / SURVEY.md 8d(ii): the reference's algorithm restated in numpy, timed the way the reference's own
ModuleTests.py:168-178 times it (time.perf_counter around a loop of .apply(chunk) calls), in two arithmetic variants
  literal3n : the reference's literal form, 3N complex fft/ifft per chunk (oracle.fftfilter_oracle.OracleLowCut/...)
  rfft2n    : overlap-save with a 2N real FFT, the arithmetic the HIP kernel performs (OracleRfft2N)
and two process layouts: one process, and one process per physical core over disjoint channels (numpy's pocketfft is
single-threaded, so cores are filled with processes - what a user of the reference with many channels would do).
"""
import multiprocessing as mp
import os
import time

import numpy as np


def physical_cores():
    """Distinct (socket, core) pairs among the CPUs this process may run on; falls back to half the logical count."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    pairs, cur = set(), {}
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if ":" in line:
                    k, v = (t.strip() for t in line.split(":", 1))
                    cur[k] = v
                elif not line.strip():
                    if cur.get("processor", "").isdigit() and int(cur["processor"]) in allowed and "core id" in cur:
                        pairs.add((cur.get("physical id", "0"), cur["core id"]))
                    cur = {}
        if cur.get("processor", "").isdigit() and int(cur["processor"]) in allowed and "core id" in cur:
            pairs.add((cur.get("physical id", "0"), cur["core id"]))
    except OSError:
        pass
    return len(pairs) if pairs else max(1, len(allowed) // 2)


def cpu_quota():
    """CPUs the container may actually use at once (cgroup v2 cpu.max / v1 cfs quota); None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def busy_cpus(interval=0.5):
    """CPUs busy right now: CPU time consumed by this container (cgroup v2 cpu.stat) or, without it, by the whole machine
    (/proc/stat) over `interval` seconds, in CPUs.  None when neither can be read."""
    def cgroup_usec():
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("usage_usec"):
                return float(line.split()[1]) * 1e-6
        raise OSError("no usage_usec")

    def proc_stat_sec():
        f = open("/proc/stat").readline().split()[1:]
        v = [float(x) for x in f]
        return (sum(v) - v[3] - (v[4] if len(v) > 4 else 0.0)) / os.sysconf("SC_CLK_TCK")  # everything but idle and iowait
    for read in (cgroup_usec, proc_stat_sec):
        try:
            a, t0 = read(), time.perf_counter()
            time.sleep(interval)
            return max(0.0, (read() - a) / (time.perf_counter() - t0))
        except (OSError, ValueError, IndexError):
            continue
    return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _make_device(filter_name, variant, n, fs, channels):
    from oracle import fftfilter_oracle as orc
    if variant == "literal3n":
        if filter_name == "lowcut":
            return orc.OracleLowCut(800, fs, n)
        if filter_name == "highcut":
            return orc.OracleHighCut(8000, fs, n)
        if filter_name == "eq3":
            return orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n)
        a, b, c = (orc.OracleLowCut(800, fs, n), orc.OracleEQ3BandFFT(100, 2, 700, -4, 8000, 5, fs, n),
                   orc.OracleHighCut(8000, fs, n))

        class _Chain:  # three devices in series, one chunk of latency each - what config 5 costs the reference
            def apply(self, x):
                return c.apply(b.apply(a.apply(x)))
        return _Chain()
    taps = {"lowcut": lambda: orc.lowcut_taps(800, fs, n), "highcut": lambda: orc.highcut_taps(8000, fs, n),
            "eq3": lambda: orc.eq3_composite_taps(100, 2, 700, -4, 8000, 5, fs, n)}.get(filter_name)
    if taps is None:
        return None  # the fused chain does not fit a 2N transform
    return orc.OracleRfft2N(taps(), n, channels=channels)


def run_worker(job):
    """(filter_name, variant, n, fs, channels_per_call, seconds, seed) -> (samples filtered, elapsed seconds)."""
    filter_name, variant, n, fs, channels, seconds, seed = job
    dev = _make_device(filter_name, variant, n, fs, channels)
    if dev is None:
        return 0, 0.0
    rng = np.random.default_rng(seed)
    shape = (n,) if variant == "literal3n" else (channels, n)
    chunks = [rng.uniform(-1, 1, shape).astype(np.float32) for _ in range(32)]
    for ch in chunks[:8]:
        dev.apply(ch)
    done = 0
    t0 = time.perf_counter()
    while True:
        for ch in chunks:
            dev.apply(ch)
        done += len(chunks)
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return done * int(np.prod(shape)), el


def measure(filter_name, n, fs, seconds_each=5.0, rfft_channels=16, max_wait_s=20.0):
    """All four figures in Msamples/s (None where a variant does not apply)."""
    cores = physical_cores()
    quota = cpu_quota()
    if quota:  # a container limited to q CPUs cannot run more than q processes at once, whatever /proc/cpuinfo lists
        cores = max(1, min(cores, int(quota + 0.999)))
    out = {"physical_cores": cores, "logical_cpus": os.cpu_count(), "cpu_model": cpu_model(), "numpy": np.__version__,
           "cgroup_cpu_quota": quota}
    # The box is shared with whatever ran before this process (the 1-minute load average remembers it for minutes): what counts is
    # how many CPUs are busy NOW.  Wait (bounded) until fewer than a quarter of the usable CPUs are; the figures say what was seen.
    try:
        out["loadavg_before"] = os.getloadavg()[0]
    except OSError:
        pass
    limit = 0.25 * cores
    waited, busy = 0.0, busy_cpus()
    while busy is not None and busy > limit and waited < max_wait_s:
        time.sleep(1.0)
        waited += 1.5
        busy = busy_cpus()
    out["busy_cpus_at_start"] = None if busy is None else round(busy, 2)
    out["waited_for_quiet_s"] = waited
    out["quiescent"] = busy is None or busy <= limit
    ctx = mp.get_context("spawn")  # the parent holds a HIP context: never fork it
    for variant, chans in (("literal3n", 1), ("rfft2n", rfft_channels)):
        samples, el = run_worker((filter_name, variant, n, fs, chans, seconds_each, 1234))
        out[f"{variant}_1proc"] = round(samples / el / 1e6, 3) if el else None
        if not el:
            out[f"{variant}_allcores"] = None
            continue
        jobs = [(filter_name, variant, n, fs, chans, seconds_each, 1234 + i) for i in range(cores)]
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            res = pool.map(run_worker, jobs)
        wall = time.perf_counter() - t0
        # every process times its own steady state; the aggregate is the sum of the per-process rates
        out[f"{variant}_allcores"] = round(sum(s / e for s, e in res if e) / 1e6, 3)
        out[f"{variant}_allcores_wall_s"] = round(wall, 2)
        if variant == "literal3n" and not out["quiescent"]:
            # measured beside other load: once more, the better of the two stands (both are reported)
            with ctx.Pool(cores) as pool:
                res = pool.map(run_worker, jobs)
            again = round(sum(s / e for s, e in res if e) / 1e6, 3)
            out["literal3n_allcores_runs"] = [out["literal3n_allcores"], again]
            out["literal3n_allcores"] = max(out["literal3n_allcores"], again)
    return out
