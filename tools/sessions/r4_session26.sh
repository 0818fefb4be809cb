#!/bin/bash
# round-4 GPU session 26: the live-session kernel after its rework (rows 0 of the twiddles in LDS, self-paired lanes in the regular
# path, no spills): kernel trace (one dispatch per session), then two PMC passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r4g_live
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4g_live/trace -o t -- python $GRAFT_REPO_ROOT/tools/live_trace.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r4g_live/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof_r4g_live/run.log | cut -c1-400
f=$(find gpurun_out/prof_r4g_live/trace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-220
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r4g_live
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
grep producer $OUT/pmc1.log | cut -c1-300
