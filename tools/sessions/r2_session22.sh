#!/bin/bash
# round-2 GPU session 22: M = 8192 as an XL plan with 32 points per thread (cross-lane pairing, three passes, half exchange)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s22; mkdir -p $O
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    d=json.loads(l); s=d.get("stream",{}); print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"| stream",s.get("value"),s.get("avg_kernel_us"))'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --steps 8 --warmup 4"
{
ADSP_PLAN_VARIANT=11 python tools/check_variant.py 8192 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//' | head -3
ADSP_PLAN_VARIANT=12 python tools/check_variant.py 4096 4 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//'
ADSP_FORCE_COMPLEX=1 ADSP_PLAN_VARIANT=11 python tools/check_variant.py 8192 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//' | head -2
for r in 1 2; do
echo "lc8192 default : $($B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "lc8192 var11   : $(ADSP_PLAN_VARIANT=11 $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "eq4096 default : $($B --no-stream-extra --filter eq3 2>>$O/err.log | line)"
echo "eq4096 var12   : $(ADSP_PLAN_VARIANT=12 $B --no-stream-extra --filter eq3 2>>$O/err.log | line)"
done
} 2>&1 | tee $O/shapes.txt
