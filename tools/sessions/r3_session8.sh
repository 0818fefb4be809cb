#!/bin/bash
# round-3 GPU session 8: batch geometry now prefers 4N at N = 1024..4096 - full suite; twiddle prefetch off for the M = 8192 plan (A/B)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s8; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l)
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"F",d["config"]["fft_size"],"kept",d["config"]["outputs_per_transform"],"cps",d["config"]["chunks_per_step"])
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-stream-extra --steps 8 --warmup 4"
{
for r in 1 2 3; do
echo "lc4096 default   : $($B 2>>$O/err.log | line)"
echo "lc4096 twpf0     : $(ADSP_LIB=abl/twpf0.so $B 2>>$O/err.log | line)"
done
echo "lc8192 default   : $($B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "lc8192 twpf0     : $(ADSP_LIB=abl/twpf0.so $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "eq4096 default   : $($B --filter eq3 2>>$O/err.log | line)"
echo "eq4096 twpf0     : $(ADSP_LIB=abl/twpf0.so $B --filter eq3 2>>$O/err.log | line)"
echo "chain default    : $($B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "chain twpf0      : $(ADSP_LIB=abl/twpf0.so $B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -25 ) > $O/pytest.log 2>&1
echo "pytest all: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head; grep -E "^[0-9.]+s " $O/pytest.log | head -5
tail -3 $O/err.log | cut -c1-300
