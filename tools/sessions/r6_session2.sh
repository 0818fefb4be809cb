#!/bin/bash
# round-6 GPU session 2: the long-kernel engine with (a) the tail-only ring update inside the multiply launch and (b) the rotated
# partition order (a channel's consecutive blocks want the same delay-line block at the same time) against round 5's kernels
# (abl/upols_r5.so = HEAD's adsp_upols.hip linked with this tree's other objects), alternating on one box; the round-6 tests; the
# multiply launch's memory counters at 1024 channels.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests_round6.txt
for r in 1 2; do
  for lib in "" abl/upols_r5.so; do
    echo "== lib=[${lib:-product}]" | tee -a $O/upols_ab.txt
    if [ -z "$lib" ]; then timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt
    else ADSP_LIB=$PWD/$lib timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_ab.txt; fi
  done
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition or smoke or twins" 2>&1 | tail -5 | tee $O/tests_long.txt
cd /tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/$O/upmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 4 --channels 1024 --block 8192 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/upols_1024ch_counters.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes over tools/bench_upols.py --only upols --calls 4 --channels 1024 --block 8192 (low cut 6 partitions and EQ 11 partitions mixed): per-dispatch averages")
for d in sorted(glob.glob("gpurun_out/r6s2/upmc_*")):
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
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/uprof -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 8 --channels 64 1024 --block 8192 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$O/uprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/upols_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/$O/uprof
