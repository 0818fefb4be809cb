#!/bin/bash
# round-5 GPU session 2: pipelined pinned staging + the fixed round-5 tests; long kernels: partitioned engine vs P-pass engines; the
# ablation builds for the speed-of-light model (tools/build_ablations.sh), headline and chain shapes; the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $O/pytest_round5.log 2>&1
echo "pytest(round5) rc=$?"; tail -6 $O/pytest_round5.log
timeout 600 python tools/bench_upols.py > $O/upols.json 2> $O/upols.err
echo "bench_upols rc=$?"; grep -v '^{"' $O/upols.json | head; tail -2 $O/upols.err
B="python bench.py --no-parity-check --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["roofline"]["avg_launch_us"], r["kernel_us_per_launch"], r["shader_mhz"])'
for m in 0 8 16 24 4 28 32 60 3 64 256 0; do
  echo "headline abl$m $(ADSP_LIB=abl/abl$m.so timeout 300 $B 2>/dev/null | python -c "$pick")" | tee -a $O/sol_headline.txt
done
for m in 0 24 4 28 32 60 3 0; do
  echo "chain abl$m $(ADSP_LIB=abl/abl$m.so timeout 300 $B --filter chain --chunk 8192 --fs 96000 2>/dev/null | python -c "$pick")" | tee -a $O/sol_chain.txt
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench(default) rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5s2/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "parity", d["max_rel_err"], "oracle", d.get("oracle_check", {}).get("max_rel_err"))
    print("numpy_api", json.dumps(d["latency"].get("numpy_api")))
except Exception as e:
    print("no line:", e)
PY
