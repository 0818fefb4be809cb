#!/bin/bash
# round-5 GPU session 4: (1) A/B of the persistent-block tuning build (a workgroup runs 2 / 4 consecutive blocks: tools/build_ablations.sh
# persist); (2) the round's profiles: rocprofv3 kernel trace + PMC passes for the headline, config 4, the chain, the per-chunk stream and the
# partitioned long-kernel engine (summaries -> profiles/r5_*).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s4
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["roofline"]["avg_launch_us"], r["kernel_us_per_launch"], r["shader_mhz"], d.get("max_rel_err"))'
for shape in "" "--fft-mult 2" "--filter chain --chunk 8192 --fs 96000"; do
  echo "[$shape] default   $(timeout 300 $B $shape 2>/dev/null | python -c "$pick")" | tee -a $O/persist_ab.txt
  for it in 1 2 4; do
    echo "[$shape] persist x$it $(ADSP_PERSIST_BUILD=1 ADSP_BLK_ITERS=$it ADSP_LIB=abl/persist.so timeout 300 $B $shape 2>/dev/null | python -c "$pick")" | tee -a $O/persist_ab.txt
  done
done
PROF_ONLY="1 2 4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5_batch > $O/prof_batch.log 2>&1; echo "profile batch rc=$?"
PROF_ONLY="1 2 4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5_chain --filter chain --chunk 8192 --fs 96000 > $O/prof_chain.log 2>&1; echo "profile chain rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5_config4 --filter highcut --channels 8192 > $O/prof_config4.log 2>&1; echo "profile config4 rc=$?"
PROF_ONLY="4 5" PROF_PASSES=5 timeout 600 bash tools/profile_gpu.sh r5_stream --mode stream --pipeline 1 > $O/prof_stream.log 2>&1; echo "profile stream rc=$?"
for t in batch chain config4 stream; do cp gpurun_out/prof_r5_$t/summary.txt $O/${t}_summary.txt 2>/dev/null; find gpurun_out/prof_r5_$t/trace -name '*kernel_stats.csv' -exec cp {} $O/${t}_kernel_stats.csv \; 2>/dev/null; done
# the partitioned long-kernel engine: kernel trace + HBM traffic of 64 channels x 88200 (Example4's shape)
cd /tmp
U="python $GRAFT_REPO_ROOT/tools/bench_upols.py --channels 64 --only upols --calls 24"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/upols_trace -o t -- $U > $GRAFT_REPO_ROOT/$O/upols_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/upols_fetch -o p -- $U > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/upols_write -o p -- $U > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r5s4/upols_profile_summary.txt
import csv, glob, collections
O = "gpurun_out/r5s4"
for f in glob.glob(O + "/upols_trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (tools/bench_upols.py --channels 64 --only upols: low cut 44099 taps, then EQ composite 88197 taps)")
    for row in list(csv.reader(open(f)))[:8]:
        print("  ", ",".join(c[:100] for c in row))
for name in ("fetch", "write"):
    for f in glob.glob(O + f"/upols_{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"][:60], row["Counter_Name"])
            if "upols" in row["Kernel_Name"]:
                acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"  {k[1]:12s} {k[0]:62s} per-dispatch avg {v / max(n, 1):14.1f} KiB (n={n})")
PY
rm -rf $O/upols_trace $O/upols_fetch $O/upols_write
cat $O/batch_summary.txt | head -40
