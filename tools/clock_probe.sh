#!/bin/bash
# run ON the GPU box: bench (batch) under a GRBM_GUI_ACTIVE pmc pass, print kernel time and effective clock
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
rm -rf /tmp/clk_$TAG
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk_$TAG -o c -- python $ROOT/bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-stream-extra "$@" > /tmp/clk_$TAG.log 2>&1
python3 - /tmp/clk_$TAG $TAG <<'PY'
import csv, glob, sys
d, tag = sys.argv[1], sys.argv[2]
cyc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fftconv" in r["Kernel_Name"]:
            cyc[r["Dispatch_Id"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fftconv" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ks = [k for k in cyc if k in dur]
if ks:
    t = sum(dur[k] for k in ks) / len(ks); c = sum(cyc[k] for k in ks) / len(ks) / 8
    print(f"{tag:14s} launches {len(ks)}  avg {t:9.1f} us  {c/1e3:9.1f} kcycles/XCD  clock {c/t/1e3:5.2f} GHz")
else:
    print(tag, "no data", open(f"/tmp/clk_{tag}.log").read()[-300:])
PY
