#!/bin/bash
# round-6 GPU session 4: the whole suite (failure names kept), block 16384 against 8192 for the low cut now that the partition order
# is rotated, and config 2 per chunk with the memory-side request counters split by destination (TCC_EA0_RDREQ_DRAM) for rings that do
# (3 slots, 192 MiB) and do not (48 slots, 3 GiB) fit the Infinity Cache.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -q -m gpu -rf 2>&1 | tail -25 | tee $O/tests_round6.txt
timeout 1500 python -m pytest tests -q -m gpu -rf 2>&1 | tail -25 | tee $O/tests_all.txt
for r in 1 2; do for b in 8192 16384; do
  echo "== block $b" | tee -a $O/upols_blocks.txt
  timeout 300 python tools/bench_upols.py --only upols --block $b 2>&1 | tail -1 | tee -a $O/upols_blocks.txt
done; done
cd /tmp
for slots in 3 48; do
  B="python $GRAFT_REPO_ROOT/bench.py --steps 768 --warmup 384 --runs 1 --no-parity-check --no-cpu-baseline --no-stream-extra --no-latency --no-configs --no-graph --mode stream --pipeline 1 --ring-slots $slots"
  for pmc in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum"; do
    tag=$(echo $pmc | tr ' ' '_' | cut -c1-20)
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${slots}_$tag -o p -- $B > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/stream_dram_requests.txt
import csv, glob, collections
print("# config 2 per chunk (4096 ch x 4096, one launch per step, one stream): memory-side requests of the L2 per dispatch, rings of 3 and 48 slots")
for d in sorted(glob.glob("gpurun_out/r6s4/pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "fftconv" not in row["Kernel_Name"]: continue
            acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"  {d.split('/')[-1]:28s} {k:28s} per-dispatch avg {v / max(n, 1):14.1f}   (n={n})")
PY
rm -rf $O/pmc_*
