#!/bin/bash
# round-4 GPU session 12: plain loads for the head and tail of a window (the part neighbouring blocks re-read): A/B on the chain, the headline, config 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s12
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-parity-check --steps 12 --warmup 4 --runs 3"
one() { # name libpath env args
  echo "$1 | $2 | $(env $3 ADSP_LIB=$2 $B $4 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["runs"]["value_msamples_s"])')"
}
for r in 1 2; do
for cfg in "chain|--filter chain --chunk 8192 --fs 96000" "headline|" "config4|--filter highcut --channels 8192"; do
  name=${cfg%%|*}; args=${cfg#*|}
  one "$name base     " abl/base.so "X=1" "$args"
  one "$name new default" "" "X=1" "$args"
  one "$name new nt-only" "" "ADSP_NT_HYBRID=0" "$args"
done; done > gpurun_out/r4s12/ab.txt 2>&1
cat gpurun_out/r4s12/ab.txt
