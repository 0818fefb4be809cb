#!/bin/bash
# round-6 GPU session 42: hunting session 39's failure - the tests whose ring steps ride a live session, 150 times over, with the HIP runtime's error log on (AMD_LOG_LEVEL=1)
# and the library's session dump (ADSP_DEBUG); the first failing run's log is kept.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s42
mkdir -p $O
fails=0
for i in $(seq 1 150); do
  AMD_LOG_LEVEL=1 ADSP_DEBUG=1 timeout 300 python -m pytest tests/test_gpu_round5.py -q -m gpu -x -rf -p no:cacheprovider -k "ride or session or pipeline or randomised" > $O/run.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp $O/run.log $O/fail_$i.log; echo "run $i rc=$rc"; fi
  if [ $fails -ge 3 ]; then break; fi
done
echo "runs=$i fails=$fails" | tee $O/summary.txt
tail -3 $O/run.log | cut -c1-200
