#!/bin/bash
# round-5 GPU session 18 (the two-halves kernel is a NEGATIVE RESULT and not in the tree; profiles/r5_upols_two_halves_in_step.txt): the multiply launch of the long-kernel engines with two output blocks per workgroup as two wave sets in step
# (ADSP_UPOLS_HALVES=2: table lines requested by both halves at about the same time) against one block per workgroup, and against the
# committed tree (build_ab/libadsp_head.so), alternating on one box; block-size policy at 256 / 512 channels; the default bench line with
# the final traffic stamps.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s18
mkdir -p $O
for r in 1 2; do
  for mode in 2 1 head; do
    echo "== halves: $mode" | tee -a $O/upols_ab.txt
    if [ $mode = head ]; then ADSP_LIB=$PWD/build_ab/libadsp_head.so timeout 300 python tools/bench_upols.py --only upols --block 8192 2>&1 | tail -1 | tee -a $O/upols_ab.txt
    else ADSP_UPOLS_HALVES=$mode timeout 300 python tools/bench_upols.py --only upols --block 8192 2>&1 | tail -1 | tee -a $O/upols_ab.txt; fi
  done
done
for b in 8192 16384; do
  echo "== block $b, 256 / 512 channels" | tee -a $O/upols_threshold.txt
  ADSP_LIB=$PWD/build_ab/libadsp_head.so timeout 300 python tools/bench_upols.py --only upols --block $b --channels 256 512 2>&1 | tail -1 | tee -a $O/upols_threshold.txt
done
ADSP_UPOLS_HALVES=2 timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition" 2>&1 | tail -5 | tee $O/tests.txt
ADSP_LIB=$PWD/build_ab/libadsp_head.so timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], [ (k, c['roofline']['traffic']) for k,c in d['configs'].items()])"
