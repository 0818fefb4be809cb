#!/bin/bash
# round-2 GPU session 16: does the power-of-two plane stride (channels x chunk x 4 B) matter?  channels 4096 vs 4104 / 4160 / 4608
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s16; mkdir -p $O
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    d=json.loads(l); print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --no-stream-extra --steps 8 --warmup 4"
{
for r in 1 2; do for c in 4096 4104 4160 4608 2048 3072; do
echo "lowcut ${c}ch x96 : $($B --channels $c 2>>$O/err.log | line)"
done; done
for c in 4096 4104; do echo "stream ${c}ch : $($B --channels $c --mode stream --steps 2048 --warmup 512 2>>$O/err.log | line)"; done
} 2>&1 | tee $O/shapes.txt
