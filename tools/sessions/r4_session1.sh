#!/bin/bash
# round-4 GPU session 1: the new tests (batch geometry at the timed size, resident launches across ring laps, launcher-free
# bench, per-rank RCCL collective, unaligned chunk sizes, live sessions), then the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s1
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --timeout 300 -p no:cacheprovider > gpurun_out/r4s1/pytest_round4.log 2>&1
echo "round4 rc=$?" ; tail -25 gpurun_out/r4s1/pytest_round4.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 120 -p no:cacheprovider -k "LC30 or HC30 or EQ30 or LC1001 or EQ1001 or LC1002 or HC1002 or EQ1002 or HC6 or LC4410" > gpurun_out/r4s1/pytest_unaligned_kat.log 2>&1
echo "kat rc=$?"; tail -5 gpurun_out/r4s1/pytest_unaligned_kat.log
timeout 300 python -m pytest tests/test_gpu_bench_contract.py -q -m gpu --timeout 280 -p no:cacheprovider > gpurun_out/r4s1/pytest_contract.log 2>&1
echo "contract rc=$?"; tail -5 gpurun_out/r4s1/pytest_contract.log
timeout 600 python bench.py > gpurun_out/r4s1/bench_default.json 2> gpurun_out/r4s1/bench_default.err
echo "bench rc=$?"; tail -c 6000 gpurun_out/r4s1/bench_default.json; tail -5 gpurun_out/r4s1/bench_default.err
