#!/bin/bash
# round-2 GPU session 8: non-temporal loads on/off for the large transforms (window overlap re-reads), FETCH_SIZE of both
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s8; mkdir -p $O
export TMPDIR=/tmp
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    try:
        d=json.loads(l); s=d.get("stream",{})
        print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"kept",d["config"]["outputs_per_transform"])
    except Exception as e: print("PARSE-FAIL",e,l[:300])'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --no-stream-extra --steps 8 --warmup 4"
bash tools/build_variant.sh nt1 -DADSP_NT=1 > $O/build.log 2>&1
{
for r in 1 2; do for lib in "" abl/nt1.so; do
echo "[$lib] chain    : $(ADSP_LIB=$lib $B --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "[$lib] lc8192   : $(ADSP_LIB=$lib $B --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "[$lib] eq4096   : $(ADSP_LIB=$lib $B --filter eq3 2>>$O/err.log | line)"
done; done
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
cd /tmp
for lib in "" abl/nt1.so; do
  ADSP_LIB=$([ -n "$lib" ] && echo $GRAFT_REPO_ROOT/$lib) rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch_$(basename "$lib" .so) -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stream-extra --no-latency --no-graph --filter chain --chunk 8192 --fs 96000 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r2s8/fetch_*")):
    acc=[0.0,0]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "fftconv" in row.get("Kernel_Name","") and row["Counter_Name"]=="FETCH_SIZE":
                acc[0]+=float(row["Counter_Value"]); acc[1]+=1
    print(d, "FETCH_SIZE KiB per dispatch", acc[0]/max(acc[1],1), "n", acc[1])
PY
