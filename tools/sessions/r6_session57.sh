#!/bin/bash
# round-6 GPU session 57: flake watch on the final library - the whole suite six times on one box (the driver's command line), every failing log kept; smoke(); the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s57
mkdir -p $O
for i in 1 2 3 4 5 6; do AMD_LOG_LEVEL=1 timeout 1200 python -m pytest tests -q -m gpu -x -rf -p no:cacheprovider > $O/all.log 2>&1; rc=$?; echo "run $i rc=$rc $(grep -E 'passed|failed' $O/all.log | tail -1)" | tee -a $O/summary.txt; if [ $rc -ne 0 ]; then cp $O/all.log $O/fail_$i.log; fi; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json
