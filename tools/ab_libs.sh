#!/bin/bash
# A/B tuning variants built by tools/build_variant.sh on the GPU box:  tools/ab_libs.sh "bench args" lib1 lib2 ...
# ("default" = the in-tree library); two alternating rounds, Msamples/s and kernel us per launch
args=$1; shift
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --steps 8 --warmup 4 $args"
for r in 1 2; do for l in "$@"; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "$l $(ADSP_LIB=$lib $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_us"])')"
done; done
