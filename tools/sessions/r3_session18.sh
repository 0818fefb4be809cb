#!/bin/bash
# round-3 GPU session 18: spectrum-stage table rows requested one group ahead (ADSP_STAGE_PREFETCH=1) - A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s18; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"| stream",s.get("value"),s.get("roofline_frac"),s.get("avg_kernel_us"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --steps 8 --warmup 4"
{
ADSP_LIB=abl/stagepf.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or batched or fused" 2>&1 | tail -2
for r in 1 2; do for lib in "" abl/stagepf.so; do
echo "[$lib] lc4096   : $(ADSP_LIB=$lib $B --no-stream-extra 2>>$O/err.log | line)"
echo "[$lib] eq4096   : $(ADSP_LIB=$lib $B --no-stream-extra --filter eq3 2>>$O/err.log | line)"
echo "[$lib] chain    : $(ADSP_LIB=$lib $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "[$lib] lc1024   : $(ADSP_LIB=$lib $B --no-stream-extra --chunk 1024 --channels 16384 2>>$O/err.log | line)"
echo "[$lib] eq512    : $(ADSP_LIB=$lib $B --filter eq3 --chunk 512 --channels 4096 2>>$O/err.log | line)"
done; done
} 2>&1 | tee $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
