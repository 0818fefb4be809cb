#!/bin/bash
# round-5 GPU session 12: the delay line in 16-byte units per lane (40 instead of 56 loads per partition and wave in the multiply kernel);
# stages requested ahead x workgroups per CU: 2 x 3, 4 x 2, 8 x 2 (-DADSP_UPOLS_AHEAD / -DADSP_UPOLS_MAC_WAVES builds under build_ab/),
# with the round's first form of the kernel (libadsp_mac_old.so) as the anchor of this box.  Then the long-kernel tests on the product build.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s12
mkdir -p $O
for r in 1 2; do
  for lib in old a2w3 a4w2 a8w2; do
    echo "== lib=[$lib]" | tee -a $O/upols_ab.txt
    ADSP_LIB=$PWD/build_ab/libadsp_mac_$lib.so timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition" 2>&1 | tail -5 | tee $O/tests.txt
