#!/bin/bash
# round-3 GPU session 5: resident ring launches without the per-workgroup cache invalidate; publication cost probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s5; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "resident or example1" 2>&1 | tail -30 ) > $O/pytest_r3.log 2>&1
echo "pytest resident: $(grep -E 'passed|failed|error' $O/pytest_r3.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_r3.log | head -20
ADSP_DEBUG=1 timeout 300 python tools/resident_probe.py 2>&1 | tee $O/probe.txt; echo "--- publications as 4-byte copies:"; ADSP_SEQ_COPY=1 timeout 300 python tools/resident_probe.py 2>&1 | tee $O/probe_copy.txt
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{}); t=s.get("two_streams",{}); r=s.get("resident",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"),"| graph",g.get("us_per_step"),"| resident",r.get("us_per_step"),r.get("kernel_us_per_step"),r.get("roofline_frac"),r.get("error"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4"
{
echo "headline+stream: $($B 2>>$O/err.log | line)"
echo "config3        : $($B --filter eq3 --chunk 512 --channels 4096 2>>$O/err.log | line)"
echo "config3 again  : $($B --filter eq3 --chunk 512 --channels 4096 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
tail -5 $O/err.log | cut -c1-300
