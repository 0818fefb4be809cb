#!/bin/bash
# round-3 GPU session 14: config 3 in resident launches on alternative M = 512 plans (variants 10, 18 - 21)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s14; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); r=s.get("resident",{})
        print("batch",d["value"],d["roofline"]["frac"],"| stream us/step",s.get("us_per_step"),"graph",s.get("graph",{}).get("us_per_step"),"| resident wall",r.get("us_per_step"),"kernel",r.get("kernel_us_per_step"),r.get("roofline_frac"),r.get("error"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --filter eq3 --chunk 512 --channels 4096 --no-cpu-baseline --no-latency --steps 8 --warmup 4"
{
for v in 10 18 19 20 21; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 512 2 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//' | head -2; done
for r in 1 2; do
echo "default : $($B 2>>$O/err.log | line)"
for v in 10 18 19 20 21; do
echo "var $v  : $(ADSP_PLAN_VARIANT=$v $B 2>>$O/err.log | line)"
done
done
} 2>&1 | tee $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
