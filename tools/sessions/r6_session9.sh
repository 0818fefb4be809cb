#!/bin/bash
# round-6 GPU session 9: two more one-line policies of the long-kernel engine against the tree (all tuning builds of the same source):
# the multiply launch at three workgroups per CU (168 VGPRs) and non-temporal stores of the forward launch's spectra.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s9
mkdir -p $O
for r in 1 2; do for l in upols_base upols_mac3 upols_znt; do
  echo "== lib=[$l]" | tee -a $O/upols_policies.txt
  ADSP_LIB=$PWD/abl/$l.so timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_policies.txt
done; done
