#!/bin/bash
# batch and stream throughput of the other shapes quoted in DESIGN.md section 5 (GPU box); one line per shape
run() { echo "$1 | $(python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 4 $2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get("stream",{}); print("batch", d["value"], "frac", d["roofline"]["frac"], "| stream", s.get("value"), "frac", s.get("roofline_frac"), "| graph", s.get("graph",{}).get("value"), s.get("graph",{}).get("us_per_step"), "| two streams", s.get("two_streams",{}).get("value"), s.get("two_streams",{}).get("us_per_step"), s.get("two_streams",{}).get("roofline_frac"))')"; }
run "lowcut N=512  x32768" "--chunk 512 --channels 32768"
run "lowcut N=1024 x16384" "--chunk 1024 --channels 16384"
run "lowcut N=2048 x8192" "--chunk 2048 --channels 8192"
run "lowcut N=8192 x2048" "--chunk 8192 --channels 2048"
run "highcut N=4096 x8192 (config 4 per GPU)" "--filter highcut --channels 8192"
run "eq3 N=512 x4096 (config 3)" "--filter eq3 --chunk 512 --channels 4096"
run "eq3 N=4096 x4096" "--filter eq3"
run "chain N=8192 x4096 96k (config 5 per GPU)" "--filter chain --chunk 8192 --fs 96000 --channels 4096 --chunks-per-step 44"
run "int16 PCM lowcut N=4096 x4096" "--io s16"
run "lowcut N=1000 x16384 (generic)" "--chunk 1000 --channels 16384"
