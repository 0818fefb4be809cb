#!/bin/bash
# round-6 GPU session 17: config 2 one launch per chunk (2N transform at N = 4096) on the two-wave 32-point plan (plans_var.hip entry 22)
# against the XL default - in round 5 the XL plan won this column by 4 % on one stream and tied library-pipelined; the 32-point plans have
# since gained 8 - 11 % in batches.  Tuning build of the final tree, alternating.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s17
mkdir -p $O
ADSP_LIB=abl/cur.so ADSP_PLAN_VARIANT=22 timeout 120 python tools/check_variant.py 4096 2>&1 | tail -4 | tee $O/ab.txt
B="python bench.py --no-cpu-baseline --no-latency --no-configs --mode stream --steps 2048 --warmup 256"
for r in 1 2 3; do for v in "" 22; do
  echo "variant=$v $(ADSP_LIB=abl/cur.so ADSP_PLAN_VARIANT=$v timeout 300 $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get("stream") or {}; print("line:", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], "| stream block:", s.get("us_per_step"), (s.get("one_stream") or {}).get("us_per_step"), (s.get("resident") or {}).get("us_per_step"))')" | tee -a $O/ab.txt
done; done
echo "== 8192 channels per chunk" | tee -a $O/ab.txt
for r in 1 2; do for v in "" 22; do
  echo "variant=$v $(ADSP_LIB=abl/cur.so ADSP_PLAN_VARIANT=$v timeout 300 $B --channels 8192 --no-stream-extra 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("line:", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])')" | tee -a $O/ab.txt
done; done
