#!/bin/bash
# round-3 GPU session 1: spectrum stage without else branches (M = 16384: 230 VGPRs, no scratch; M = 8192: 144 VGPRs);
# M = 8192 at four workgroups per CU (variants 11 / 12); chain profile with FETCH/WRITE counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s1; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --steps 8 --warmup 4"
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
{
ADSP_PLAN_VARIANT=11 python tools/check_variant.py 8192 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//' | head -4
ADSP_PLAN_VARIANT=12 python tools/check_variant.py 4096 4 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//' | head -4
for r in 1 2; do
echo "headline       : $($B 2>>$O/err.log | line)"
echo "chain          : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "lc8192 default : $($B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "lc8192 var11   : $(ADSP_PLAN_VARIANT=11 $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "eq4096 default : $($B --no-stream-extra --filter eq3 2>>$O/err.log | line)"
echo "eq4096 var12   : $(ADSP_PLAN_VARIANT=12 $B --no-stream-extra --filter eq3 2>>$O/err.log | line)"
done
echo "chain stream   : $($B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
PROF_PASSES=5 bash tools/profile_gpu.sh r3_chain --filter chain --chunk 8192 --fs 96000 > $O/prof.log 2>&1
tail -25 gpurun_out/prof_r3_chain/summary.txt
tail -3 $O/err.log | cut -c1-300
