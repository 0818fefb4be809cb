#!/bin/bash
# round-5 GPU session 10: the chain (config 5) on the 32-points-per-thread x 512-thread plan (variant 13) now that it, too, loads the head and
# tail of its windows with plain loads - against the default 64-point plan, alternating on one box; PMC traffic of both.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s10
mkdir -p $O
run() { if [ -z "$1" ]; then env -u ADSP_PLAN_VARIANT "${@:2}"; else env ADSP_PLAN_VARIANT=$1 "${@:2}"; fi; }
C="python bench.py --no-cpu-baseline --no-stream-extra --no-latency --no-configs --steps 8 --warmup 2 --runs 3 --filter chain --chunk 8192 --fs 96000"
pickb='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["runs"]; print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], r["kernel_us_per_launch"], r["shader_mhz"], d.get("max_rel_err"))'
for r in 1 2; do for v in "" 13; do
  echo "chain variant=[$v] $(run "$v" timeout 300 $C 2>/dev/null | python -c "$pickb")" | tee -a $O/chain_ab.txt
done; done
echo "chain variant=[13] all loads non-temporal $(ADSP_NT_HYBRID=0 run 13 timeout 300 $C 2>/dev/null | python -c "$pickb")" | tee -a $O/chain_ab.txt
ADSP_PLAN_VARIANT=13 timeout 300 python tools/check_variant.py 8192 2>&1 | grep -v "^$" | tail -4 | tee -a $O/chain_ab.txt
cd /tmp
for pmc in FETCH_SIZE WRITE_SIZE; do
  ADSP_PLAN_VARIANT=13 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --runs 1 --no-parity-check --no-cpu-baseline --no-stream-extra --no-latency --no-configs --filter chain --chunk 8192 --fs 96000 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee -a gpurun_out/r5s10/chain_ab.txt
import csv, glob, collections
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/r5s10/pmc_{name}/**/*counter_collection.csv", recursive=True):
        acc = [0.0, 0]
        for row in csv.DictReader(open(f)):
            if "fftconv" in row["Kernel_Name"]:
                acc[0] += float(row["Counter_Value"]); acc[1] += 1
        print(f"variant 13 {name} per-dispatch avg {acc[0] / max(acc[1], 1):.1f} KiB (n={acc[1]})")
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
