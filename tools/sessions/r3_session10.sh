#!/bin/bash
# round-3 GPU session 10: the float64 flavour of the kernels (sample_format s16_f64) - tests, throughput; float flavour unchanged?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s10; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "exact_fft" 2>&1 | tail -30 ) > $O/pytest_f64.log 2>&1
echo "pytest f64: $(grep -E 'passed|failed|error' $O/pytest_f64.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|assert|differ" $O/pytest_f64.log | head -20
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"F",d["config"]["fft_size"],"kept",d["config"]["outputs_per_transform"],"| stream",s.get("value"),s.get("roofline_frac"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-stream-extra --steps 8 --warmup 4"
{
echo "f32 lc4096        : $($B 2>>$O/err.log | line)"
echo "s16 lc4096        : $($B --io s16 2>>$O/err.log | line)"
echo "s16_f64 lc4096    : $($B --io s16_f64 2>>$O/err.log | line)"
echo "s16_f64 lc4096 2N : $($B --io s16_f64 --fft-mult 2 2>>$O/err.log | line)"
echo "s16_f64 lc1024    : $($B --io s16_f64 --chunk 1024 --channels 16384 2>>$O/err.log | line)"
echo "s16_f64 eq512     : $($B --io s16_f64 --filter eq3 --chunk 512 --channels 4096 2>>$O/err.log | line)"
echo "s16_f64 stream    : $($B --io s16_f64 --mode stream --steps 1024 --warmup 256 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest all: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
tail -3 $O/err.log | cut -c1-300
