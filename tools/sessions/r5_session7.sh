#!/bin/bash
# (the first run of this session selected variant 0 where it meant "default": an EMPTY ADSP_PLAN_VARIANT did that - fixed in the library since)
# round-5 GPU session 7: config 2's per-chunk pattern measured the way the bench line measures it (stream_figures: wall clock without per-launch
# events, then kernel time; one stream and library-pipelined), two-wave plan (default) against the XL plan (variant 26), alternating on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s7
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-latency --no-configs --no-parity-check --steps 4 --warmup 1 --runs 1 --prewarm-ms 100"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["stream"]; o=s["one_stream"]; print("pipelined", s.get("us_per_step"), s.get("runs_us_per_step"), "one stream", o["us_per_step"], "kernel", o["avg_kernel_us"], "graph", o.get("graph", {}).get("us_per_step"), "resident", s.get("resident", {}).get("us_per_step"), s.get("resident", {}).get("kernel_us_per_step"))'
for r in 1 2; do for v in "" 26; do
  echo "variant=[$v] $(ADSP_PLAN_VARIANT=$v timeout 300 $B 2>/dev/null | python -c "$pick")" | tee -a $O/stream_ab.txt
done; done
# the same for 8192 channels per GPU (config 4's channel count, per chunk)
for v in "" 26; do
  echo "8192ch variant=[$v] $(ADSP_PLAN_VARIANT=$v timeout 300 $B --channels 8192 --chunks-per-step 49 2>/dev/null | python -c "$pick")" | tee -a $O/stream_ab.txt
done
