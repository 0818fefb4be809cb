#!/bin/bash
# round-3 GPU session 7: batch mode on 4N transforms (M = 8192 plan at four workgroups per CU keeps 3.5 N of 4 N for the cut filters)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s7; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l)
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"F",d["config"]["fft_size"],"kept",d["config"]["outputs_per_transform"],"cps",d["config"]["chunks_per_step"])
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-stream-extra --steps 8 --warmup 4"
{
for r in 1 2; do
echo "lc4096 2N        : $($B 2>>$O/err.log | line)"
echo "lc4096 4N        : $($B --fft-mult 4 2>>$O/err.log | line)"
echo "hc4096x8192 2N   : $($B --filter highcut --channels 8192 2>>$O/err.log | line)"
echo "hc4096x8192 4N   : $($B --filter highcut --channels 8192 --fft-mult 4 2>>$O/err.log | line)"
done
echo "lc2048x8192 2N   : $($B --chunk 2048 --channels 8192 2>>$O/err.log | line)"
echo "lc2048x8192 4N   : $($B --chunk 2048 --channels 8192 --fft-mult 4 2>>$O/err.log | line)"
echo "lc1024x16384 2N  : $($B --chunk 1024 --channels 16384 2>>$O/err.log | line)"
echo "lc1024x16384 4N  : $($B --chunk 1024 --channels 16384 --fft-mult 4 2>>$O/err.log | line)"
echo "lc512x32768 2N   : $($B --chunk 512 --channels 32768 2>>$O/err.log | line)"
echo "lc512x32768 4N   : $($B --chunk 512 --channels 32768 --fft-mult 4 2>>$O/err.log | line)"
echo "eq4096 (4N)      : $($B --filter eq3 2>>$O/err.log | line)"
echo "s16 lc4096 2N    : $($B --io s16 2>>$O/err.log | line)"
echo "s16 lc4096 4N    : $($B --io s16 --fft-mult 4 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
