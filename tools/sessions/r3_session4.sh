#!/bin/bash
# round-3 GPU session 4: resident ring launches (tests + bench figures), full suite, M = 3072 at three waves per SIMD (variant 14)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s4; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{}); t=s.get("two_streams",{}); r=s.get("resident",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"F",d["config"]["fft_size"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"),"| graph",g.get("us_per_step"),"| two",t.get("us_per_step"),t.get("roofline_frac"),"| resident",r.get("us_per_step"),r.get("kernel_us_per_step"),r.get("roofline_frac"),r.get("error"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
( timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "resident" 2>&1 | tail -30 ) > $O/pytest_r3.log 2>&1
echo "pytest resident: $(grep -E 'passed|failed|error' $O/pytest_r3.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_r3.log | head -20
{
echo "headline+stream: $($B 2>>$O/err.log | line)"
echo "config3        : $($B --filter eq3 --chunk 512 --channels 4096 2>>$O/err.log | line)"
echo "hc8192 1.5N    : $($B --mode stream --steps 2048 --warmup 512 --filter highcut --channels 8192 --fft-mult 1.5 2>>$O/err.log | line)"
echo "hc8192 1.5N v14: $(ADSP_PLAN_VARIANT=14 $B --mode stream --steps 2048 --warmup 512 --filter highcut --channels 8192 --fft-mult 1.5 2>>$O/err.log | line)"
echo "lc4096 1.5N v14: $(ADSP_PLAN_VARIANT=14 $B --mode stream --steps 2048 --warmup 512 --fft-mult 1.5 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest all: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 3000 $O/bench_default.json
tail -5 $O/err.log | cut -c1-300
