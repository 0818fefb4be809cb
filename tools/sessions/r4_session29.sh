#!/bin/bash
# round-4 GPU session 29 (NOT kept: -2 % for the four-wave plan, session 30): the self-paired butterflies of EVERY XL plan through the regular pair operations (M = 4096: config 2's
# per-chunk kernel): whole suite, then stream mode (one stream and library-pipelined) against the tree before
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s29
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider -x > gpurun_out/r4s29/pytest_gpu.log 2>&1
echo "suite rc=$?"; grep -n "passed\|failed" gpurun_out/r4s29/pytest_gpu.log | tail -2
for r in 1 2; do for l in head selfxl; do for pipe in 1 2; do
  echo "$l pipeline $pipe $(ADSP_LIB=abl/$l.so python bench.py --mode stream --pipeline $pipe --runs 3 --no-parity-check --no-cpu-baseline --no-stream-extra --no-latency 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["runs"]["value_msamples_s"])')"
done; done; done 2>&1 | tee gpurun_out/r4s29/stream_ab.txt
