#!/bin/bash
# round-2 GPU session 5: new default plans for M = 8192 / 16384 (half exchange), radix-32 two-level twiddles A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s5; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --steps 8 --warmup 4"
bash tools/build_variant.sh tw32off -DADSP_TW2_RADIX32=0 > $O/build.log 2>&1 &
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
wait
{
for r in 1 2; do for lib in "" abl/tw32off.so; do
echo "[$lib] headline : $(ADSP_LIB=$lib $B 2>>$O/err.log | line)"
echo "[$lib] lc8192   : $(ADSP_LIB=$lib $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "[$lib] eq4096   : $(ADSP_LIB=$lib $B --filter eq3 2>>$O/err.log | line)"
echo "[$lib] chain    : $(ADSP_LIB=$lib $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
done; done
echo "chain stream : $($B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain round-1 plan (variant 4): $(ADSP_PLAN_VARIANT=4 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
