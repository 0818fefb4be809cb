#!/bin/bash
# round-2 GPU session 14: twiddle prefetch before the exchange also for passes with two butterflies per thread (M = 8192, 16384)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s14; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l)
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"])
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --no-stream-extra --steps 8 --warmup 4"
bash tools/build_variant.sh pre2 -DADSP_TW_PREFETCH_NB=2 > $O/build.log 2>&1
{
for r in 1 2; do for lib in "" abl/pre2.so; do
echo "[$lib] lc8192   : $(ADSP_LIB=$lib $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "[$lib] eq4096   : $(ADSP_LIB=$lib $B --filter eq3 2>>$O/err.log | line)"
echo "[$lib] chain    : $(ADSP_LIB=$lib $B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "[$lib] lc2048   : $(ADSP_LIB=$lib $B --chunk 2048 --channels 8192 2>>$O/err.log | line)"
echo "[$lib] eq512    : $(ADSP_LIB=$lib $B --filter eq3 --chunk 512 2>>$O/err.log | line)"
done; done
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
