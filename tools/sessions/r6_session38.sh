#!/bin/bash
# round-6 GPU session 38: a stage's successor requested AFTER its multiply-adds (a scheduling barrier between them: the loads land in the registers just consumed; hoisted
# above the multiply-adds - what the compiler does by itself - they need a third set).  ra_a2 = two stages ahead for both block sizes (122 registers, four waves per SIMD);
# ra_m3a4 = blocks of 8192 with FOUR stages ahead (140 registers, three per CU), 16384 with two; default = 8192: two ahead, hoisted (148 registers), 16384: one ahead.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s38
mkdir -p $O
for r in 1 2; do for l in default ra_a2 ra_m3a4; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "== lib=[$l]" | tee -a $O/ab.txt
  for args in "--channels 64 256 1024 --block 8192" "--channels 64 256 1024 --block 16384"; do
  ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols $args 2>/dev/null | tail -1 | tee -a $O/ab.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200','').replace('lowcut_44099','lc').replace('eq3_88197','eq'):(v['upols']['us_per_call'],v['upols']['block']) for k,v in d.items()})"
  done
done; done
