#!/bin/bash
# round-6 GPU session 54: the bench line's long-kernel block also at 1024 channels (time only) - its test, the default line and the driver's arguments (size, time), the bench contract tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s54
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_bench_contract.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -p no:cacheprovider -k "bench" 2>&1 | tail -3
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt; wc -c $O/bench_default.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err ) 2> $O/bench_time2.txt; echo "bench rc=$?"; tail -3 $O/bench_time2.txt
python -c "
import json
for f in ('bench_default','bench_driver_args'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); lk=d['latency']['long_kernels']
    print(f, d['value'], d['roofline']['frac'], {k:(v['us_per_call'],v['roofline_frac']) for k,v in lk['at_1024_channels'].items()}, lk['lowcut_44099_taps']['us_per_call'], lk['eq3_88197_taps']['us_per_call'])
"
tail -c 2000 $O/bench_driver_args.json | head -c 400
