#!/bin/bash
# round-6 GPU session 33: whose gain was it?  old3a2 = the UNSPLIT form of rounds 5 - 6a (adsp_upols.hip of commit ad0babb) with three multiply workgroups per CU and two stages
# ahead (134 registers, no scratch) - alternating against that form as it shipped (presplit: two per CU, four ahead) and the split form (default: three per CU, two ahead).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s33
mkdir -p $O
for r in 1 2; do for l in presplit old3a2 default; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "== lib=[$l]" | tee -a $O/ab.txt
  ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols 2>/dev/null | tail -1 | tee -a $O/ab.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200',''):(v['upols']['us_per_call'],v['upols']['block']) for k,v in d.items()})"
  ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols --block 8192 --channels 256 1024 2>/dev/null | tail -1 | tee -a $O/ab.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200',''):(v['upols']['us_per_call'],v['upols']['block']) for k,v in d.items()})"
done; done
