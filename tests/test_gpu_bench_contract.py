"""The driver's contract with bench.py, checked on the GPU box with a tiny workload: exactly one JSON line, last on stdout,
with the metric / roofline / cpu_baseline objects the driver and the judge read.  Run with -m gpu on MI355X."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--channels", "256", "--chunks-per-step", "12", "--prewarm-ms", "20",
           "--cpu-seconds", "0.2", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={**os.environ, **(env or {})})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip()]
    return json.loads(lines[-1]), lines  # the JSON line is the LAST line (RCCL's banner is flushed before it)


def test_default_line_has_everything_the_driver_reads():
    d, lines = run_bench("--steps", "3", "--warmup", "1", "--no-configs")  # (configs 4 / 5 in the line: tests/test_gpu_round5.py)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches"] == 3 and r["algorithmic_bytes_per_launch"] == 8 * d["config"]["samples_per_step"]
    # value = samples per step x steps / wall time: consistent with ms_per_step
    assert abs(d["value"] - d["config"]["samples_per_step"] / d["ms_per_step"] / 1e3) <= 0.01 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Msamples/s" and c["cores"] >= 1 and c["value"] > 0 and len(c["variants_msamples_s"]) == 4
    assert d["stream"]["value"] > 0 and d["latency"]["numpy_api_apply_us_per_call"] > 0
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1  # exactly one JSON line
    # SURVEY 8d statistics (round 4): five timed regions of exactly K steps, `value` is the median run, all runs are listed
    rr = d["runs"]
    assert rr["n"] == 5 and rr["statistic"] == "median" and len(rr["value_msamples_s"]) == 5 and len(rr["ms_per_step"]) == 5
    assert rr["min"] <= rr["median"] <= rr["max"] and abs(rr["median"] - d["value"]) <= 1e-3 * d["value"]
    assert sorted(rr["value_msamples_s"])[2] == rr["median"] and len(rr["shader_mhz"]) == 5
    assert all(m is None or 500.0 < m < 3000.0 for m in rr["shader_mhz"]) and any(m is not None for m in rr["shader_mhz"])
    # the timed output is parity-checked after the timed region (float64 direct sum on the GPU, 1e-5 like every parity test)
    assert d["parity_checked"] is True and 0.0 <= d["max_rel_err"] <= 1e-5
    assert d["parity"]["channels"] >= 16 and d["parity"]["samples"] >= 16 * 12 * 4096 and d["parity"]["launch"] >= 1


def test_other_workloads_and_the_rccl_path_print_the_same_line():
    d, _ = run_bench("--steps", "2", "--warmup", "1", "--filter", "chain", "--chunk", "8192", "--fs", "96000", "--channels", "64",
                     "--chunks-per-step", "11", "--no-stream-extra", "--no-latency", "--no-cpu-baseline")
    assert d["config"]["kernel_taps"] == 9401 and d["config"]["outputs_per_transform"] == 22528 and d["value"] > 0
    d, _ = run_bench("--steps", "64", "--warmup", "8", "--mode", "stream", "--no-latency", "--no-cpu-baseline")
    assert d["config"]["mode"] == "stream" and d["config"]["chunks_per_step"] == 1 and d["roofline"]["launches"] == 64
    d, lines = run_bench("--steps", "2", "--warmup", "1", "--no-stream-extra", "--no-latency", "--no-cpu-baseline", "--no-configs",
                         env={"ADSP_BENCH_FORCE_PG": "1", "MASTER_PORT": "29541"})  # init_process_group("nccl") at world size 1
    assert d["n_gpus"] == 1 and d["value"] > 0 and lines[-1].lstrip().startswith("{")
    # ... and the same collective once more through the C ABI (adsp_bcast_spectrum_rank), cross-checked against the torch carrier
    chk = d["abi_carrier_check"]
    assert chk["ranks_ok"] == 1 and chk["equal_on_all_ranks"] and chk["matches_torch_carrier"], chk
