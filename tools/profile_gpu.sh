#!/bin/bash
# Run ON the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py; results -> gpurun_out/prof_<tag>/
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# a step is one launch over a resident batch (bench.py): 6 timed launches in batch mode, 768 in stream mode
case " $* " in *" stream "*) STEPS="--steps 768 --warmup 384";; *) STEPS="--steps 6 --warmup 2";; esac
BENCH="python $ROOT/bench.py $STEPS --runs 1 --no-parity-check --no-cpu-baseline --no-stream-extra --no-latency --no-configs --no-graph $*"
PASSES=${PROF_PASSES:-8}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE32_INSTS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  [ $i -gt $PASSES ] && break
  if [ -n "${PROF_ONLY:-}" ]; then case " $PROF_ONLY " in *" $i "*) ;; *) continue;; esac; fi  # e.g. PROF_ONLY="4 5": the HBM traffic passes only
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
summ = open(os.path.join(out, "summary.txt"), "w")
def emit(*a):
    print(*a); print(*a, file=summ)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    emit("== kernel stats", os.path.relpath(f, out))
    for row in list(csv.reader(open(f)))[:6]:
        emit("  ", ",".join(c[:110] for c in row))
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "fftconv" not in row.get("Kernel_Name", ""): continue
            k = row["Counter_Name"]; acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            emit(f"  {k:32s} per-dispatch avg {v / max(n,1):16.1f}   (n={n})")
summ.close()
PY
