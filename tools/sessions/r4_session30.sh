#!/bin/bash
# round-4 GPU session 30: stream mode (768 timed steps after 384 of warm-up, three regions), tree before vs the generalised XL stage (kernel us per launch is the figure to read: the top-level --mode stream wall clock of --pipeline 2 is not - bench.py's "stream" object is)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s30
for r in 1 2; do for l in ${LIBS:-head selfxl}; do for pipe in 1 2; do
  echo "$l pipeline $pipe $(ADSP_LIB=abl/$l.so python bench.py --mode stream --pipeline $pipe --steps 768 --warmup 384 --runs 3 --no-parity-check --no-cpu-baseline --no-stream-extra --no-latency --no-graph 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "us/step", round(d["ms_per_step"]*1e3,2), "kernel us", d["roofline"]["avg_launch_us"], d["runs"]["value_msamples_s"])')"
done; done; done 2>&1 | tee gpurun_out/r4s30/stream_ab.txt
