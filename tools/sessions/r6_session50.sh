#!/bin/bash
# round-6 GPU session 50 (final tree: split spectra, thread 0 parked, blocks of 16384 on 512 threads, the straddling block computed once): the whole -m gpu suite, smoke(), the default bench line
# and the driver's arguments, the long-kernel engines with their kernel trace and cache counters.  (The filter kernels' sources - and the traffic stamps 744b78a09c4fad37 - are unchanged.)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s50
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo "bench(default) rc=$?"; tail -3 $O/bench_time.txt; wc -c $O/bench_default.json; tail -c 1200 $O/bench_default.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench(driver args) rc=$?"
timeout 600 python tools/bench_upols.py > $O/upols_bench.log 2>&1; tail -1 $O/upols_bench.log > $O/upols_bench.json; echo "bench_upols rc=$?"; cut -c1-1800 $O/upols_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/uprof -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 8 --channels 64 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$O/uprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/upols_64ch_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/$O/uprof
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/$O/upmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 4 --channels 1024 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/upols_1024ch_counters.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes over tools/bench_upols.py --only upols --calls 4 --channels 1024 (blocks of 16384 on 512 threads for both kernels - low cut 3 partitions, EQ 6 - mixed): per-dispatch averages")
for d in sorted(glob.glob("gpurun_out/r6s50/upmc_*")):
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
timeout 600 python examples/harness_timing.py > $O/harness_timing.json 2> $O/harness_timing.err; echo "harness rc=$?"
