#!/bin/bash
# round-5 GPU session 22: last sweep of the multiply launch on 16-byte table entries - eight stages ahead (248 VGPRs) and two stages ahead
# at three workgroups per CU (168 VGPRs + 36 B) against the product (four ahead, two per CU); then the whole -m gpu suite on the final tree.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s22
mkdir -p $O
for r in 1 2; do
  for lib in product a8 a2w3; do
    echo "== $lib" | tee -a $O/upols_ab.txt
    if [ $lib = product ]; then timeout 300 python tools/bench_upols.py --only upols --block 8192 2>&1 | tail -1 | tee -a $O/upols_ab.txt
    else ADSP_LIB=$PWD/build_ab/libadsp_$lib.so timeout 300 python tools/bench_upols.py --only upols --block 8192 2>&1 | tail -1 | tee -a $O/upols_ab.txt; fi
  done
done
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2; grep -E "^FAILED|Error" $O/pytest_all.log | head -5
