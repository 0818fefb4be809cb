#!/bin/bash
# round-5 GPU session 3: ring steps riding a live session (pipeline depth 3); host staging variants; the ablations that write no output;
# the whole suite on the new kernels; the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "ride or refused or staging" > $O/pytest_new.log 2>&1
echo "pytest(new) rc=$?"; tail -12 $O/pytest_new.log
B="python bench.py --no-parity-check --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["roofline"]["avg_launch_us"], r["kernel_us_per_launch"], r["shader_mhz"])'
for m in 0 16 24 28 60 8; do
  echo "headline abl$m $(ADSP_BENCH_NO_SANITY=1 ADSP_LIB=abl/abl$m.so timeout 300 $B 2>/dev/null | python -c "$pick")" | tee -a $O/sol_headline.txt
done
for m in 0 16 24 28 60; do
  echo "chain abl$m $(ADSP_BENCH_NO_SANITY=1 ADSP_LIB=abl/abl$m.so timeout 300 $B --filter chain --chunk 8192 --fs 96000 2>/dev/null | python -c "$pick")" | tee -a $O/sol_chain.txt
done
python - <<'PY' > gpurun_out/r5s3/host_staging.txt 2>&1
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyaudiodsptools_amd import FirEngine, design, synth
n, C, steps = 4096, 4096, 16
fir = design.FirStream(design.lowcut_kernel(800, 44100, n), n)
xd = torch.empty((steps, C, n), device="cuda")
synth.fill_device(xd, 4321, 0, 0, C, n, steps)
x = xd.cpu().numpy(); del xd
out = np.empty_like(x)
for label, env in (("pinned, 4 copy threads", {}), ("pinned, 8 copy threads", {"ADSP_HOST_COPY_THREADS": "8"}), ("pinned, 2 copy threads", {"ADSP_HOST_COPY_THREADS": "2"}),
                   ("direct slabs (runtime stages)", {"ADSP_HOST_STAGING": "direct"}), ("one piece (round 4)", {"ADSP_HOST_UNPIPELINED": "1"})):
    for k, v in env.items(): os.environ[k] = v
    eng = FirEngine(fir, channels=C, optimize_for="batch")
    ts = []
    for i in range(4):
        eng.reset(); t0 = time.perf_counter(); eng.apply_host(x, out=out); ts.append(time.perf_counter() - t0)
    eng.close()
    for k in env: del os.environ[k]
    print(f"{label:32s} {[round(t * 1e3, 1) for t in ts]} ms  -> {x.nbytes / min(ts[1:]) / 1e9:.1f} GB/s each direction", flush=True)
PY
cat $O/host_staging.txt
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_round5.py > $O/pytest_all.log 2>&1
echo "pytest(all other gpu) rc=$?"; tail -6 $O/pytest_all.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench(default) rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5s3/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "parity", d["max_rel_err"])
    print("stream live_pipeline", json.dumps(d["stream"].get("live_pipeline"))[:300])
    c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]
    print("config3", {k: c3.get(k) for k in ("us_per_step", "value", "roofline_frac", "launch_per_step")})
    print("config3 live_pipeline", json.dumps(c3.get("live_pipeline"))[:400])
except Exception as e:
    print("no line:", e)
PY
