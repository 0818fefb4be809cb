#!/bin/bash
# round-2 GPU session 7: generalised paired pass (radix-16 pairing at 64 / 32 points per thread)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s7; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --steps 8 --warmup 4"
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
{
ADSP_PLAN_VARIANT=12 python tools/check_variant.py 8192 4 2>&1 | grep -E "variant|Error|error" | sed -E 's/plan=\{[^}]*\}//'
ADSP_PLAN_VARIANT=14 python tools/check_variant.py 8192 2>&1 | grep -E "variant|Error|error" | sed -E 's/plan=\{[^}]*\}//'
ADSP_FORCE_COMPLEX=1 ADSP_PLAN_VARIANT=12 python tools/check_variant.py 8192 4 2>&1 | grep -E "variant|Error|error" | sed -E 's/plan=\{[^}]*\}//' | head -2
for r in 1 2; do
echo "chain default : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain var12   : $(ADSP_PLAN_VARIANT=12 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "lc8192 default: $($B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "lc8192 var14  : $(ADSP_PLAN_VARIANT=14 $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
done
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
