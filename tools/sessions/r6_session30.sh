#!/bin/bash
# round-6 GPU session 30: the split-spectrum multiply launch at THREE workgroups per CU (ADSP_UPOLS_MAC_WAVES=3: 64 channels x 88200 are 704 workgroups - 512 places at two
# per CU, 768 at three), with four and with two stages of requests ahead; alternating against the library before the change and the split form at two per CU.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s30
mkdir -p $O
for r in 1 2; do for l in presplit default mac3 mac3a2; do
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
