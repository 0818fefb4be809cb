#!/bin/bash
# round-6 GPU session 56: the last binary of the round (comment-only rebuild of adsp_upols.o) - the whole suite (the driver's command), smoke(), the bench with the driver's arguments
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s56
mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt; python -c "
import json; d=json.loads(open('$O/bench_driver_args.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], {k:v['roofline']['frac'] for k,v in d['configs'].items()})"
