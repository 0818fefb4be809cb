#!/bin/bash
# round-6 GPU session 35: the split / re-packing with thread 0's registers PARKED in LDS instead of an if / else over all registers (the else's inputs stayed alive across the
# then: +64 registers - the source of every spill and of the 148 registers of the 8192-point multiply launch): no scratch anywhere, 122 - 127 registers.
#   default   = 8192: three multiply workgroups per CU, two stages ahead; 16384: 64 points per thread in 256 threads (as before)
#   b16k512a1 = 16384 on 32 points per thread in 512 threads (125 / 122 registers, no scratch), ONE stage ahead everywhere
#   m4a1      = the same + the 8192-point multiply launch at four workgroups per CU (122 registers)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s35
mkdir -p $O
for l in default m4a1; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "== parity lib=[$l]" | tee -a $O/pytest_subset.txt
  ADSP_LIB=$lib timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_moduletests.py tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee -a $O/pytest_subset.txt
done
for r in 1 2; do for l in presplit default b16k512a1 m4a1; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "== lib=[$l]" | tee -a $O/ab.txt
  for args in "--channels 64 256 1024 --block 8192" "--channels 256 1024 --block 16384"; do
  ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols $args 2>/dev/null | tail -1 | tee -a $O/ab.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200','').replace('lowcut_44099','lc').replace('eq3_88197','eq'):(v['upols']['us_per_call'],v['upols']['block']) for k,v in d.items()})"
  done
done; done
