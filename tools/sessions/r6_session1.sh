#!/bin/bash
# round-6 GPU session 1: the call-pattern tests (Example4's in-place loop, non-finite samples, the callback thread), the whole GPU suite on
# the round's first tree, the default bench line and the long-kernel baseline of this box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s1
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests_round6.txt
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/tests_all.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee $O/upols_baseline.txt
