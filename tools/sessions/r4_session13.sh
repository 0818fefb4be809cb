#!/bin/bash
# round-4 GPU session 13: chain (config 5) build variants - load fence off, twiddle prefetch off
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s13
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-parity-check --steps 12 --warmup 4 --runs 3"
one() { echo "$1 | $(ADSP_LIB=$2 $B $3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["runs"]["value_msamples_s"])')"; }
for r in 1 2; do
  for l in "" abl/nofence.so abl/notwpf.so; do
    one "chain ${l:-default}" "$l" "--filter chain --chunk 8192 --fs 96000"
  done
  for l in "" abl/nofence.so; do
    one "headline ${l:-default}" "$l" ""
  done
done > gpurun_out/r4s13/ab.txt 2>&1
cat gpurun_out/r4s13/ab.txt
