#!/bin/bash
# round-4 GPU session 27: PMC passes over live sessions fed by HOST stores only (under --pmc rocprofv3 serialises dispatches, so a
# publishing kernel cannot run beside the session: the passes of sessions 18 and 26 mostly counted the polling of starved sessions)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp LIVE_PRODUCERS=host,host
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r4g_live_host
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/live_trace.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/live_trace.py > $OUT/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY' | tee $OUT/pmc_summary.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "fftconv_live" not in row.get("Kernel_Name", ""): continue
            k = row["Counter_Name"]; acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"  {k:28s} per-dispatch avg {v / max(n,1):18.1f}   (n={n})")
PY
grep -h producer $OUT/pmc1.log $OUT/pmc2.log | cut -c1-300
grep fftconv_live $OUT/pmc1/*kernel_trace.csv | awk -F, '{print "dispatch ns", $(NF-11)-$(NF-12)}' | tr -d '"'
