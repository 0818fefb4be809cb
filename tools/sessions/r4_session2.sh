#!/bin/bash
# round-4 GPU session 2: live sessions after the rewrite (buffer builtins, fetch-ahead, wide relay sweep), library-pipelined
# ring steps, the bench line with the new stream / live figures; load-mode A/B of the live kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s2
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --timeout 300 -p no:cacheprovider -k "live or pipelined" > gpurun_out/r4s2/pytest_live.log 2>&1
echo "live rc=$?" ; tail -30 gpurun_out/r4s2/pytest_live.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4s2/bench_default.json 2> gpurun_out/r4s2/bench_default.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s2/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["frac"], d["runs"]["value_msamples_s"])
print("stream", {k: d["stream"].get(k) for k in ("value", "us_per_step", "roofline_frac", "runs_us_per_step", "pipelined_error")})
print("one_stream", {k: d["stream"].get("one_stream", {}).get(k) for k in ("value", "us_per_step", "avg_kernel_us", "roofline_frac")})
c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
print("config3", {k: c3.get(k) for k in ("us_per_step", "avg_kernel_us")}, "pipelined", c3.get("pipelined"))
print("resident", c3.get("resident"))
print("live", json.dumps(c3.get("resident_live"), indent=1))
PY
tail -3 gpurun_out/r4s2/bench_default.err
python - <<'PY' > gpurun_out/r4s2/live_modes.txt 2>&1
import argparse, json, sys, torch
sys.path.insert(0, ".")
import bench
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
dev = torch.device("cuda", 0)
for mode in (2, 1, 0):
    r = bench.live_figures(a3, bench.make_fir(a3), dev, 8, 4096, 512, steps=4096, load_mode=mode)
    print("load_mode", mode, json.dumps({k: r[k] for k in ("stream_producer", "host_producer", "round_trip_us")}))
for ch in (2048, 1024):
    r = bench.live_figures(a3, bench.make_fir(a3), dev, 8, ch, 512, steps=4096)
    print("channels", ch, json.dumps({k: r[k] for k in ("stream_producer", "host_producer", "round_trip_us")}))
PY
cat gpurun_out/r4s2/live_modes.txt
