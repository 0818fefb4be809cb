#!/bin/bash
# round-6 GPU session 10: the last tree - whole suite, smoke, the default bench line and the driver's.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s10
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed|FAILED" $O/pytest_all.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo "bench(default) rc=$?"; tail -3 $O/bench_time.txt; wc -c $O/bench_default.json; tail -c 2000 $O/bench_default.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench(driver args) rc=$?"; head -c 700 $O/bench_driver_args.json
timeout 300 python bench.py --explain --no-cpu-baseline --no-configs --steps 4 --warmup 2 --runs 1 2> $O/explain.txt > /dev/null; wc -l $O/explain.txt
