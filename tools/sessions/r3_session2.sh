#!/bin/bash
# round-3 GPU session 2: new tests (bcast, multi-stream ring ordering, bench N=2 on one GPU, single-process mode); chain A/Bs:
# variant 13 (32 points per thread, 512 threads, two workgroups of eight waves), radix-32 twiddles from the table (tw32off)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s2; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{}); t=s.get("two_streams",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"),"| graph",g.get("us_per_step"),"| two",t.get("us_per_step"),t.get("roofline_frac"),t.get("graph",{}).get("us_per_step"), t.get("error"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
( timeout 1200 python -m pytest tests/test_gpu_round3.py -m gpu -q -x 2>&1 | tail -30 ) > $O/pytest_r3.log 2>&1
echo "pytest round3: $(grep -E 'passed|failed|error' $O/pytest_r3.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_r3.log | head -20
( timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_round3.py 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
{
ADSP_PLAN_VARIANT=13 python tools/check_variant.py 8192 4 2>&1 | grep -E "variant|Error" | sed -E 's/plan=\{[^}]*\}//' | head -4
for r in 1 2; do
echo "chain default  : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain var13    : $(ADSP_PLAN_VARIANT=13 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain tw32off  : $(ADSP_LIB=abl/tw32off.so $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "lc8192 default : $($B --no-stream-extra --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "lc8192 tw32off : $(ADSP_LIB=abl/tw32off.so $B --no-stream-extra --chunk 8192 --channels 2048 2>>$O/err.log | line)"
done
echo "headline+stream: $($B 2>>$O/err.log | line)"
echo "config3 stream : $($B --filter eq3 --chunk 512 --channels 4096 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
tail -5 $O/err.log | cut -c1-300
