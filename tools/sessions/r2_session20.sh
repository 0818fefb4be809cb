#!/bin/bash
# round-2 GPU session 20: history hand-over inside the kernel vs the side-stream device copy (ADSP_HOST_RING_COPY=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s20; mkdir -p $O
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    d=json.loads(l); print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"ms/step",d["ms_per_step"])'; }
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
B="python bench.py --no-cpu-baseline --no-latency --no-graph --no-stream-extra --steps 12 --warmup 4"
{
for r in 1 2 3; do
echo "in-kernel hand-over : $($B 2>>$O/err.log | line)"
echo "side-stream copy    : $(ADSP_HOST_RING_COPY=1 $B 2>>$O/err.log | line)"
done
echo "in-kernel  N=512 x32768: $($B --chunk 512 --channels 32768 2>>$O/err.log | line)"
echo "side copy  N=512 x32768: $(ADSP_HOST_RING_COPY=1 $B --chunk 512 --channels 32768 2>>$O/err.log | line)"
echo "in-kernel  chain: $($B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "side copy  chain: $(ADSP_HOST_RING_COPY=1 $B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "in-kernel  s16: $($B --io s16 2>>$O/err.log | line)"
echo "side copy  s16: $(ADSP_HOST_RING_COPY=1 $B --io s16 2>>$O/err.log | line)"
} 2>&1 | tee $O/shapes.txt
