#!/bin/bash
# round-6 GPU session 34: blocks of 16384 on 32 points per thread in 512 threads with the split-spectrum multiply launch (128 registers, four waves per SIMD; 60 B of scratch with two
# stages ahead - b16k512 - and with one - b16k512a1) against the default (64 points per thread in 256 threads, two waves per SIMD), 1024 channels x 88200.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s34
mkdir -p $O
for r in 1 2; do for l in default b16k512 b16k512a1; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "== lib=[$l]" | tee -a $O/ab.txt
  ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols --block 16384 --channels 256 1024 2>/dev/null | tail -1 | tee -a $O/ab.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200',''):(v['upols']['us_per_call'],v['upols']['block']) for k,v in d.items()})"
done; done
