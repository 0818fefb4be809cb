#!/bin/bash
# round-6 GPU session 31: the split-spectrum long-kernel engine with its new defaults (three multiply workgroups per CU, two stages ahead) - parity subset, then alternating
# against the library before the change and against the forward launch held to 128 registers (four workgroups per CU, 8 bytes of scratch: fwd4).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s31
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_moduletests.py tests/test_gpu_fuzz.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_subset.txt
for r in 1 2; do for l in presplit default fwd4; do
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
