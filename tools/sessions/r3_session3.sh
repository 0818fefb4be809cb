#!/bin/bash
# round-3 GPU session 3: the 3 * 2^k plan (M = 3072, F = 1.5 N) - tests, then stream-mode A/B against the 2N transform
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s3; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{}); t=s.get("two_streams",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"F",d["config"]["fft_size"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"),"| graph",g.get("us_per_step"),"| two",t.get("us_per_step"),t.get("roofline_frac"),t.get("graph",{}).get("us_per_step"), t.get("error"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
( timeout 1200 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "three_times or bcast" 2>&1 | tail -30 ) > $O/pytest_r3.log 2>&1
echo "pytest M3072: $(grep -E 'passed|failed|error' $O/pytest_r3.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_r3.log | head -20
{
for r in 1 2; do
echo "stream 1.5N : $($B --mode stream --steps 2048 --warmup 512 2>>$O/err.log | line)"
echo "stream 2N   : $($B --mode stream --steps 2048 --warmup 512 --fft-mult 2 2>>$O/err.log | line)"
echo "hc8192ch 1.5N: $($B --mode stream --steps 2048 --warmup 512 --filter highcut --channels 8192 2>>$O/err.log | line)"
echo "hc8192ch 2N  : $($B --mode stream --steps 2048 --warmup 512 --filter highcut --channels 8192 --fft-mult 2 2>>$O/err.log | line)"
done
echo "headline+stream: $($B 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest all: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
tail -5 $O/err.log | cut -c1-300
