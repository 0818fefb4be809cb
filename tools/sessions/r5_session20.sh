#!/bin/bash
# round-5 GPU session 20 (final tree after the long-kernel engines' table rework): the whole -m gpu suite, the default bench line, the
# long-kernel engines beside the P-pass engines, their kernel trace at 64 channels and cache counters at 1024.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s20
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo "bench(default) rc=$?"; tail -3 $O/bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5s20/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "parity", d["max_rel_err"], "oracle", d["oracle_check"]["max_rel_err"])
for k, c in d.get("configs", {}).items(): print("  ", k, c.get("value"), c["roofline"]["frac"], c["roofline"]["traffic"])
s = d["stream"]; print("  stream", s.get("us_per_step"), s.get("roofline_frac"), "one", s["one_stream"]["us_per_step"])
c3 = d["latency"]["config3_eq3_2048_stereo_pairs_x_512"]; print("  config3", c3.get("us_per_step"), c3.get("launch_per_step", {}).get("us_per_step"))
print("  long_kernels", json.dumps(d["latency"]["long_kernels"])[:700])
print("  host", d["latency"]["numpy_api"]["apply_host_1gib"]["gb_per_s_each_direction"], "cpu", d["cpu_baseline"]["value"])
PY
timeout 600 python tools/bench_upols.py > $O/upols_bench.log 2>&1; tail -1 $O/upols_bench.log > $O/upols_bench.json; echo "bench_upols rc=$?"; cut -c1-1600 $O/upols_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/uprof -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 8 --channels 64 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$O/uprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/upols_64ch_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/$O/uprof
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/$O/upmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 4 --channels 1024 --block 8192 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r5s20/upols_1024ch_counters.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes over tools/bench_upols.py --only upols --calls 4 --channels 1024 --block 8192 (low cut 6 partitions and EQ 11 partitions mixed): per-dispatch averages")
for d in sorted(glob.glob("gpurun_out/r5s20/upmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "upols" not in kn: continue
            k = ("forward " if "forward" in kn else "multiply ") + row["Counter_Name"]
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"  {k:44s} per-dispatch avg {v / max(n, 1):18.1f}   (n={n})")
PY
rm -rf $O/upmc_*
