#!/bin/bash
# round-2 GPU session 11: two-pass plan for config 3's stream transform (M = 512 = 32 x 16, 32 points per thread)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s11; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"| stream",s.get("value"),s.get("us_per_step"),s.get("avg_kernel_us"),"| graph",g.get("value"),g.get("us_per_step"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
{
for v in 18 19; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 512 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//'; done
for r in 1 2; do
echo "cfg3 default : $($B --filter eq3 --chunk 512 2>>$O/err.log | line)"
for v in 18 19; do echo "cfg3 var$v : $(ADSP_PLAN_VARIANT=$v $B --filter eq3 --chunk 512 2>>$O/err.log | line)"; done
echo "lc512x32768 default : $($B --chunk 512 --channels 32768 2>>$O/err.log | line)"
for v in 18 19; do echo "lc512x32768 var$v : $(ADSP_PLAN_VARIANT=$v $B --chunk 512 --channels 32768 2>>$O/err.log | line)"; done
done
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
