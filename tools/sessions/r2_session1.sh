#!/bin/bash
# round-2 GPU session 1: full -m gpu suite, default bench, chain / large-M A/B, profiles.  Run via gpurun from the repo root.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s1; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"),"| graph",g.get("value"),g.get("us_per_step"),g.get("error"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest done: $(tail -1 $O/pytest.log)"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default: $(line < $O/bench_default.json)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driverargs.json 2>> $O/bench_default.err; echo "driver-args: $(line < $O/bench_driverargs.json)"
{
echo "chain trimmed   : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain untrimmed : $($B --no-stream-extra --filter chain --chunk 8192 --fs 96000 --trim 0 2>>$O/err.log | line)"
echo "chain trim var2 : $(ADSP_PLAN_VARIANT=2 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain trim var3 : $(ADSP_PLAN_VARIANT=3 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain trim var1 : $(ADSP_PLAN_VARIANT=1 $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain + stream  : $($B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "lowcut N=8192 x2048 : $($B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "eq3 N=4096 x4096    : $($B --filter eq3 2>>$O/err.log | line)"
echo "eq3 N=512 x4096 cfg3: $($B --filter eq3 --chunk 512 2>>$O/err.log | line)"
echo "highcut x8192 cfg4  : $($B --filter highcut --channels 8192 --chunks-per-step 48 2>>$O/err.log | line)"
echo "lowcut FORCE_PG nccl: $(ADSP_BENCH_FORCE_PG=1 $B --no-stream-extra 2>>$O/err.log | line)"
echo "stream ring4        : $($B --mode stream --ring-slots 4 --steps 2048 --warmup 512 2>>$O/err.log | line)"
echo "stream ring3        : $($B --mode stream --steps 2048 --warmup 512 2>>$O/err.log | line)"
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
PROF_PASSES=5 bash tools/profile_gpu.sh r2_chain --filter chain --chunk 8192 --fs 96000 > $O/prof_chain.log 2>&1
PROF_PASSES=5 bash tools/profile_gpu.sh r2_stream --mode stream > $O/prof_stream.log 2>&1
tail -40 gpurun_out/prof_r2_chain/summary.txt
tail -30 gpurun_out/prof_r2_stream/summary.txt
