#!/bin/bash
# round-6 GPU session 36: blocks of 16384 now run on 512 threads - where does design.choose_uniform_block's threshold belong?  Both block sizes at 64 ... 1024 channels
# (chunk 88200, both kernels), twice; then the long-kernel tests on the new default.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s36
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_moduletests.py tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/pytest_subset.txt
for r in 1 2; do for b in 8192 16384; do
  echo "== block $b" | tee -a $O/blocks.txt
  timeout 600 python tools/bench_upols.py --only upols --channels 64 128 256 512 1024 --block $b 2>/dev/null | tail -1 | tee -a $O/blocks.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200','').replace('lowcut_44099','lc').replace('eq3_88197','eq'):v['upols']['us_per_call'] for k,v in d.items()})"
done; done
