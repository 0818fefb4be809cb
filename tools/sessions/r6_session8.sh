#!/bin/bash
# round-6 GPU session 8: the long-kernel forward launch with plain window loads (every sample of a window is read by two workgroups)
# against the non-temporal loads of rounds 5 - 6 (abl/upols_nt0.so against abl/upols_nt1.so, same tuning build otherwise), alternating,
# with the forward launch's fetched bytes; the round's new tests; WavBank.process with parallel transpositions and a kept engine.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s8
mkdir -p $O
for r in 1 2; do for l in upols_nt1 upols_nt0; do
  echo "== lib=[$l]" | tee -a $O/upols_forward_loads.txt
  ADSP_LIB=$PWD/abl/$l.so timeout 300 python tools/bench_upols.py --only upols 2>&1 | tail -1 | tee -a $O/upols_forward_loads.txt
done; done
cd /tmp
for l in upols_nt1 upols_nt0; do
  ADSP_LIB=$GRAFT_REPO_ROOT/abl/$l.so timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$l -o p -- python $GRAFT_REPO_ROOT/tools/bench_upols.py --only upols --calls 4 --channels 1024 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee -a $O/upols_forward_loads.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r6s8/pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "upols" not in kn: continue
            k = ("forward " if "forward" in kn else "multiply ") + row["Counter_Name"]
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"  {d.split('/')[-1]:16s} {k:32s} per-dispatch avg {v / max(n, 1):14.1f} KiB  (n={n})   [1024 channels, both kernels mixed]")
PY
rm -rf $O/pmc_*
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_pcm16.py -q -m gpu -rf 2>&1 | tail -6 | tee $O/tests.txt
python - <<'PY' | tee $O/wavbank.txt
import sys, json, torch
sys.path.insert(0, ".")
import bench
r = bench.host_batch_figures(torch.device("cuda", 0))
print(json.dumps({k: v for k, v in r["wavbank_process"].items() if k != "note"}))
f = bench.long_kernel_figures(torch.device("cuda", 0))
print({k: v["us_per_call"] for k, v in f.items() if isinstance(v, dict) and "us_per_call" in v})
PY
