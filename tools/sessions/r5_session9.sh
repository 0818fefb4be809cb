#!/bin/bash
# round-5 GPU session 9: M = 2048 as ONE wave per transform (plan variants 28 - 30) against the default two-wave plan: N = 2048 per chunk
# (8192 channels: the samples per step of config 2), N = 1024 batches; and how N = 2048 per chunk compares with N = 4096 per chunk at equal
# samples per step (the premise of running config 2's step as two half-chunk transforms).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s9
mkdir -p $O
for v in 28 30; do ADSP_PLAN_VARIANT=$v timeout 300 python tools/check_variant.py 2048 2>&1 | grep "stream" | tee -a $O/check_variants.txt; done
ADSP_PLAN_VARIANT=29 timeout 300 python tools/check_variant.py 1024 2>&1 | grep "batch" | tee -a $O/check_variants.txt
B="python bench.py --no-cpu-baseline --no-latency --no-configs --no-parity-check --steps 4 --warmup 1 --runs 1 --prewarm-ms 100"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["stream"]; o=s["one_stream"]; print("pipelined", s.get("us_per_step"), s.get("runs_us_per_step"), "one stream", o["us_per_step"], "kernel", o["avg_kernel_us"], "resident", s.get("resident", {}).get("us_per_step"))'
run() { if [ -z "$1" ]; then env -u ADSP_PLAN_VARIANT "${@:2}"; else env ADSP_PLAN_VARIANT=$1 "${@:2}"; fi; }
for r in 1 2; do
  echo "N=4096 x 4096ch default    $(run "" timeout 300 $B 2>/dev/null | python -c "$pick")" | tee -a $O/stream_ab.txt
  for v in "" 28 30; do
    echo "N=2048 x 8192ch variant=[$v] $(run "$v" timeout 300 $B --chunk 2048 --channels 8192 2>/dev/null | python -c "$pick")" | tee -a $O/stream_ab.txt
  done
done
pickb='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], r["shader_mhz"], d.get("max_rel_err"))'
BB="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3 --chunk 1024 --channels 16384"
for v in "" 29 ""; do echo "N=1024 batch variant=[$v] $(run "$v" timeout 300 $BB 2>/dev/null | python -c "$pickb")" | tee -a $O/stream_ab.txt; done
