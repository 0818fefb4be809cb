#!/bin/bash
# round-2 GPU session 2: suite with the exact mode, small-M XL variants, stream stagger, RCCL world-1 bench, int16 histogram
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s2; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"),"| graph",g.get("value"),g.get("us_per_step"),g.get("error"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -10
python tools/pcm16_histogram.py > $O/pcm16_histogram.json 2> $O/pcm16.err; cat $O/pcm16_histogram.json | tr -d '\n' | cut -c1-1500; echo
{
for v in 4 5; do ADSP_PLAN_VARIANT=$v python tools/check_variant.py 512 2>&1 | grep variant; done
ADSP_PLAN_VARIANT=6 python tools/check_variant.py 1024 2>&1 | grep variant
echo "cfg3 default : $($B --filter eq3 --chunk 512 2>>$O/err.log | line)"
echo "cfg3 var4    : $(ADSP_PLAN_VARIANT=4 $B --filter eq3 --chunk 512 2>>$O/err.log | line)"
echo "cfg3 var5    : $(ADSP_PLAN_VARIANT=5 $B --filter eq3 --chunk 512 2>>$O/err.log | line)"
echo "lc512x32768 default : $($B --chunk 512 --channels 32768 2>>$O/err.log | line)"
echo "lc512x32768 var4    : $(ADSP_PLAN_VARIANT=4 $B --chunk 512 --channels 32768 2>>$O/err.log | line)"
echo "lc512x32768 var5    : $(ADSP_PLAN_VARIANT=5 $B --chunk 512 --channels 32768 2>>$O/err.log | line)"
echo "lc1024x16384 default: $($B --chunk 1024 --channels 16384 2>>$O/err.log | line)"
echo "lc1024x16384 var6   : $(ADSP_PLAN_VARIANT=6 $B --chunk 1024 --channels 16384 2>>$O/err.log | line)"
for st in 0 2 4 8; do echo "stream stagger $st : $(ADSP_STAGGER=$st $B --mode stream --steps 2048 --warmup 512 --no-graph 2>>$O/err.log | line)"; done
echo "lowcut FORCE_PG nccl: $(ADSP_BENCH_FORCE_PG=1 $B --no-stream-extra 2>>$O/err.log | line)"
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
tail -5 $O/err.log | cut -c1-300
