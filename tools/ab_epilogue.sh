#!/bin/bash
# cost of the fused output effects on the headline workload (run on the GPU box): Msamples/s and kernel us per launch
B="python bench.py --no-cpu-baseline --no-stream-extra"
for m in batch stream; do
  for e in none volume softclip harddist saturator tremolo; do
    echo "$m $e $($B --mode $m --effect $e 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"])')"
  done
done
