#!/bin/bash
# round-6 GPU session 37: the block-size question below 64 channels (1 ... 32 channels x 88200, both kernels, both blocks, twice)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s37
mkdir -p $O
for r in 1 2; do for b in 8192 16384; do
  echo "== block $b" | tee -a $O/blocks.txt
  timeout 600 python tools/bench_upols.py --only upols --channels 1 4 8 16 32 --block $b 2>/dev/null | tail -1 | tee -a $O/blocks.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k.replace('_taps','').replace('_x_88200','').replace('lowcut_44099','lc').replace('eq3_88197','eq'):v['upols']['us_per_call'] for k,v in d.items()})"
done; done
