#!/bin/bash
# round-2 GPU session 3: full suite, default bench (driver contract), N>1 control flow on one GPU, RCCL world-1, headline profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s3; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{}); g=s.get("graph",{})
        print("value",d["value"],"n_gpus",d["n_gpus"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"traffic",d["roofline"]["traffic"],"| stream",s.get("value"),s.get("roofline_frac"),"| graph",g.get("value"),g.get("us_per_step"))
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) nproc: $(nproc) loadavg: $(cat /proc/loadavg)"
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default: $(line < $O/bench_default.json)"
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('cpu_baseline'))); print(json.dumps(d.get('latency')))"
echo "FORCE_PG nccl : $(ADSP_BENCH_FORCE_PG=1 python bench.py --no-cpu-baseline --no-latency --no-stream-extra --steps 8 --warmup 4 2>>$O/err.log | line)"
echo "2 ranks, one GPU, gloo: $(ADSP_BENCH_SINGLE_DEVICE=1 ADSP_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --channels 1024 2>>$O/err.log | line)"
bash tools/profile_gpu.sh r2_batch > $O/prof_batch.log 2>&1
cat gpurun_out/prof_r2_batch/summary.txt | head -60
tail -3 $O/err.log | cut -c1-300
