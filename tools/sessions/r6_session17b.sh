#!/bin/bash
# round-6 GPU session 17b: the same A/B through the default line's `stream` block (library-pipelined depth 2, one stream, resident launches)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s17
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-latency --no-configs --steps 8 --warmup 4"
echo "== default line's stream block: pipelined us/step, one stream us/step (kernel us), resident us/step" | tee -a $O/ab.txt
for r in 1 2 3; do for v in "" 22; do
  echo "variant=$v $(ADSP_LIB=abl/cur.so ADSP_PLAN_VARIANT=$v timeout 300 $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get("stream") or {}; o=s.get("one_stream") or {}; print(s.get("us_per_step"), s.get("runs_us_per_step"), o.get("us_per_step"), o.get("avg_kernel_us"), (s.get("resident") or {}).get("us_per_step"))')" | tee -a $O/ab.txt
done; done
