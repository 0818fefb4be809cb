#!/bin/bash
# round-3 GPU session 16: rocprofv3 kernel trace of resident ring launches against per-step launches (config 3's shape)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r3s16; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --filter eq3 --chunk 512 --channels 4096 --no-cpu-baseline --no-latency --steps 4 --warmup 2 > $O/bench.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fftconv" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
    by[(r["Kernel_Name"][:95], grid)].append(d)
with open(out + "/summary.txt", "w") as fh:
    for (name, grid), ds in sorted(by.items(), key=lambda kv: -len(kv[1])):
        line = f"{len(ds):6d} dispatches  grid {grid:9d}  avg {sum(ds)/len(ds):10.2f} us  min {min(ds):10.2f}  max {max(ds):10.2f}  {name}"
        print(line); fh.write(line + "\n")
PY
tail -2 $O/bench.log | cut -c1-400
