#!/bin/bash
# round-2 GPU session 4: half-buffer LDS exchange variants (M = 8192 at 3 workgroups per CU, 64 points per thread plans)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s4; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --steps 8 --warmup 4"
{
for v in 4 6 8 12; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 8192 2>&1 | grep -E "variant|Error|error" | grep -v batch.*F=32768; done
for v in 5 9; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 4096 4 2>&1 | grep -E "variant|Error|error"; done
for v in 7 13; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 8192 4 2>&1 | grep -E "variant|Error|error"; done
for v in 10 11; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 4096 2>&1 | grep -E "variant|Error|error" | grep -v F=16384; done
for r in 1 2; do
echo "lc8192 default : $($B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
for v in 4 12; do echo "lc8192 var$v    : $(ADSP_PLAN_VARIANT=$v $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"; done
echo "eq4096 default : $($B --filter eq3 2>>$O/err.log | line)"
for v in 5 9; do echo "eq4096 var$v    : $(ADSP_PLAN_VARIANT=$v $B --filter eq3 2>>$O/err.log | line)"; done
echo "chain default  : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain var7     : $(ADSP_PLAN_VARIANT=7 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain var13    : $(ADSP_PLAN_VARIANT=13 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "headline default: $($B 2>>$O/err.log | line)"
echo "headline var10  : $(ADSP_PLAN_VARIANT=10 $B 2>>$O/err.log | line)"
echo "headline var11  : $(ADSP_PLAN_VARIANT=11 $B 2>>$O/err.log | line)"
done
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
