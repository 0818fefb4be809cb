#!/bin/bash
# A/B an environment switch of the library on the GPU box:  tools/ab_env.sh "bench args" VAR=value ...   ("-" = nothing set)
args=$1; shift
B="python bench.py --no-cpu-baseline --no-stream-extra $args"
for r in 1 2; do for kv in "$@"; do
  if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi
  echo "$kv $(env $pre $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"], d["config"].get("spectrum"))')"
done; done
