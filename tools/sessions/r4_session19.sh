#!/bin/bash
# round-4 GPU session 19: live session, table pointers laundered as global pointers (global_load instead of flat_load) vs the
# tree before, plus three ablations of the new build (tables from two addresses / butterflies as copies / no LDS exchange)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s19
for v in live_base live_glob live_glob_abl3 live_glob_abl32 live_glob_abl4 live_base live_glob; do
ADSP_LIB=$PWD/abl/$v.so ADSP_LIVE_TRACE=3000 ADSP_LIVE_TRACE_WG=7 timeout 200 python - 2>&1 <<'PY' | grep -v amdgpu.ids | cut -c1-600
import json, os, sys, torch
sys.path.insert(0, ".")
import bench
a3 = bench.parse(["--filter", "eq3", "--chunk", "512", "--fs", "44100", "--channels", "4096"])
try:
    r = bench.live_figures(a3, bench.make_fir(a3), torch.device("cuda", 0), 8, 4096, 512, steps=4096, prewarm_ms=30.0)
    sys.stderr.flush()
    print(os.path.basename(os.environ["ADSP_LIB"]), json.dumps({k: r[k] for k in ("stream_producer", "host_producer", "round_trip_us") if k in r}))
except Exception as e:
    print(os.path.basename(os.environ["ADSP_LIB"]), "FAILED", repr(e)[:300])
PY
done > gpurun_out/r4s19/variants.txt 2>&1
grep "us_per_step\|trace\|FAILED" gpurun_out/r4s19/variants.txt | cut -c1-330
