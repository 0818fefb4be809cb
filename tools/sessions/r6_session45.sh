#!/bin/bash
# round-6 GPU session 45: the block that straddles a call boundary computed ONCE (its second part carried to the next call's output by the per-channel workgroups) -
# the long-kernel / parity / fuzz tests, then alternating against the library before the change (abl/precarry.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s45
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_moduletests.py tests/test_gpu_fuzz.py tests/test_gpu_effects.py tests/test_gpu_pcm16.py -q -m gpu -rf -p no:cacheprovider > $O/pytest_subset.log 2>&1; echo "rc=$?"; tail -12 $O/pytest_subset.log | cut -c1-300
for r in 1 2; do for l in precarry default; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "== lib=[$l]" | tee -a $O/ab.txt
  for args in "--channels 64 256 1024" "--channels 16 --block 8192"; do
  ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols $args 2>/dev/null | tail -1 | tee -a $O/ab.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200','').replace('lowcut_44099','lc').replace('eq3_88197','eq'):(v['upols']['us_per_call'],v['upols']['block']) for k,v in d.items()})"
  done
done; done
