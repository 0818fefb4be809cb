#!/bin/bash
# round-6 GPU session 20 (final tree of the round: session 16's kernels + the ModuleTests fixtures; the abl16384 switch of session 18 changed the source stamp, not the product code): the whole -m gpu suite, smoke(), the default bench line (and the driver's arguments), the long-kernel
# engines with their kernel trace and cache counters, and the PMC passes for the traffic stamps of the four bench shapes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s20
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf > $O/pytest_all.log 2>&1
echo "pytest(all gpu) rc=$?"; grep -E "passed|failed" $O/pytest_all.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo "bench(default) rc=$?"; tail -3 $O/bench_time.txt; wc -c $O/bench_default.json; tail -c 2100 $O/bench_default.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench(driver args) rc=$?"
timeout 600 python tools/bench_upols.py > $O/upols_bench.log 2>&1; tail -1 $O/upols_bench.log > $O/upols_bench.json; echo "bench_upols rc=$?"; cut -c1-1800 $O/upols_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/uprof -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 8 --channels 64 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$O/uprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/upols_64ch_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/$O/uprof
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/$O/upmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 4 --channels 1024 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/upols_1024ch_counters.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes over tools/bench_upols.py --only upols --calls 4 --channels 1024 (default blocks: 16384 for both kernels - low cut 3 partitions, EQ 6 - mixed): per-dispatch averages")
for d in sorted(glob.glob("gpurun_out/r6s20/upmc_*")):
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
PROF_ONLY="1 2 4 5" PROF_PASSES=5 timeout 900 bash tools/profile_gpu.sh r6_batch > $O/prof_batch.log 2>&1; echo "profile batch rc=$?"
PROF_ONLY="1 2 4 5" PROF_PASSES=5 timeout 900 bash tools/profile_gpu.sh r6_chain --filter chain --chunk 8192 --fs 96000 > $O/prof_chain.log 2>&1; echo "profile chain rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r6_config4 --filter highcut --channels 8192 > $O/prof_config4.log 2>&1; echo "profile config4 rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r6_stream --mode stream --pipeline 1 > $O/prof_stream.log 2>&1; echo "profile stream rc=$?"
for t in batch chain config4 stream; do cp gpurun_out/prof_r6_$t/summary.txt $O/${t}_summary.txt 2>/dev/null; find gpurun_out/prof_r6_$t/trace -name '*kernel_stats.csv' -exec cp {} $O/${t}_kernel_stats.csv \; 2>/dev/null; done
grep -E "fftconv|FETCH|WRITE" $O/*_summary.txt | cut -c1-200
# soak: the whole GPU suite twice more on the same box (order-independent state, leaks between tests)
for r in 2 3; do timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | tee -a $O/pytest_soak.txt; done
