#!/bin/bash
# round-3 GPU session 11: M = 3072 in four passes with 24 points per thread (two waves per transform): variants 15 / 16 / 17
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s11; mkdir -p $O
export TMPDIR=/tmp
for v in 15 16 17; do
ADSP_PLAN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "three_times and f32" 2>&1 | tail -3 | sed "s/^/variant $v: /"
done
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l)
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"F",d["config"]["fft_size"],"kept",d["config"]["outputs_per_transform"])
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-stream-extra --mode stream --steps 2048 --warmup 512"
{
for r in 1 2; do
echo "lc4096 stream 2N      : $($B 2>>$O/err.log | line)"
echo "lc4096 stream 1.5N    : $($B --fft-mult 1.5 2>>$O/err.log | line)"
for v in 15 16 17; do
echo "lc4096 stream 1.5N v$v: $(ADSP_PLAN_VARIANT=$v $B --fft-mult 1.5 2>>$O/err.log | line)"
done
done
echo "hc8192ch stream 2N     : $($B --filter highcut --channels 8192 2>>$O/err.log | line)"
for v in 15 16 17; do
echo "hc8192ch stream 1.5N v$v: $(ADSP_PLAN_VARIANT=$v $B --filter highcut --channels 8192 --fft-mult 1.5 2>>$O/err.log | line)"
done
} 2>&1 | tee $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
