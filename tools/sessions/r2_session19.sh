#!/bin/bash
# round-2 GPU session 19: stream mode with consecutive steps on two HIP streams in turn
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s19; mkdir -p $O
show() { python -c 'import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get("stream",{})
print({k:s.get(k) for k in ("value","us_per_step","avg_kernel_us","roofline_frac")}, "graph", s.get("graph",{}).get("us_per_step"), "two_streams", s.get("two_streams"))'; }
B="python bench.py --no-cpu-baseline --no-latency --steps 4 --warmup 2"
{
echo "cfg2 : $($B 2>>$O/err.log | show)"
echo "cfg3 : $($B --filter eq3 --chunk 512 2>>$O/err.log | show)"
echo "cfg4 : $($B --filter highcut --channels 8192 2>>$O/err.log | show)"
echo "n8192: $($B --chunk 8192 --channels 2048 2>>$O/err.log | show)"
for ns in; do echo "cfg3 --streams $ns: $(python bench.py --no-cpu-baseline --no-latency --mode stream --filter eq3 --chunk 512 --steps 8192 --warmup 2048 --streams $ns --no-graph 2>>$O/err.log | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"])')"; done
for ns in; do echo "cfg2 --streams $ns: $(python bench.py --no-cpu-baseline --no-latency --mode stream --steps 2048 --warmup 512 --streams $ns --no-graph 2>>$O/err.log | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"])')"; done
} 2>&1 | tee $O/shapes.txt
tail -3 $O/err.log | cut -c1-300
