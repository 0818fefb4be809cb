#!/bin/bash
# round-2 GPU session 10: full suite; channels per workgroup for config 3's stream transform; fused effect on the 64-point chain kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s10; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"| stream",s.get("value"),s.get("us_per_step"),s.get("avg_kernel_us"),"| graph",g.get("value"),g.get("us_per_step"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
bash tools/build_variant.sh oldbig '-DADSP_PLAN_16384=Plan<16384,32,3,32,32,16,1>' > $O/build.log 2>&1 &
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
wait
{
for r in 1 2; do
echo "cfg3 default (2 ch/WG): $($B --filter eq3 --chunk 512 2>>$O/err.log | line)"
for v in 15 16 17; do echo "cfg3 var$v : $(ADSP_PLAN_VARIANT=$v $B --filter eq3 --chunk 512 2>>$O/err.log | line)"; done
echo "chain+softclip default : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 --effect softclip 2>>$O/err.log | line)"
echo "chain+softclip old plan: $(ADSP_LIB=abl/oldbig.so $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 --effect softclip 2>>$O/err.log | line)"
echo "chain s16 default : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 --io s16 2>>$O/err.log | line)"
echo "chain s16 old plan: $(ADSP_LIB=abl/oldbig.so $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 --io s16 2>>$O/err.log | line)"
done
for v in 15 16 17; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 512 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//' | head -2; done
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
