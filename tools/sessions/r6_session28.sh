#!/bin/bash
# round-6 GPU session 28: the tree as it stands after sessions 21 - 27 (pinned host windows, compressor / gate across the lanes, the long-kernel engine's
# host window): the whole -m gpu suite, smoke(), the default bench line and the driver's arguments, the harness example.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s28
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo "bench(default) rc=$?"; tail -3 $O/bench_time.txt; wc -c $O/bench_default.json; tail -c 2100 $O/bench_default.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err ) 2> $O/bench_driver_time.txt; echo "bench(driver args) rc=$?"; tail -3 $O/bench_driver_time.txt
timeout 600 python examples/harness_timing.py > $O/harness_timing.json 2> $O/harness_timing.err; echo "harness rc=$?"; tail -c 600 $O/harness_timing.json
