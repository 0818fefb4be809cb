#!/bin/bash
# round-2 GPU session 15: config 4's per-GPU shape (8192 ch x 4096) against the headline shape on one box, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s15; mkdir -p $O
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    d=json.loads(l); print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"launches",d["roofline"]["launches"])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --no-stream-extra"
{
for r in 1 2 3; do
echo "lowcut 4096ch x96 : $($B --steps 8 --warmup 4 2>>$O/err.log | line)"
echo "highcut 8192ch x48: $($B --steps 8 --warmup 4 --filter highcut --channels 8192 --chunks-per-step 48 2>>$O/err.log | line)"
echo "highcut 4096ch x96: $($B --steps 8 --warmup 4 --filter highcut 2>>$O/err.log | line)"
echo "lowcut 8192ch x48 : $($B --steps 8 --warmup 4 --channels 8192 --chunks-per-step 48 2>>$O/err.log | line)"
echo "highcut 8192ch x96: $($B --steps 8 --warmup 4 --filter highcut --channels 8192 --chunks-per-step 96 2>>$O/err.log | line)"
done
} 2>&1 | tee $O/shapes.txt
