#!/bin/bash
# round-2 GPU session 21: wave priority (s_setprio) raised while a workgroup issues its window loads / stores
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2s21; mkdir -p $O
line() { python -c 'import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    d=json.loads(l); s=d.get("stream",{}); print("value",d["value"],"frac",d["roofline"]["frac"],"us/launch",d["roofline"]["avg_launch_us"],"| stream",s.get("value"),s.get("avg_kernel_us"))'; }
B="python bench.py --no-cpu-baseline --no-latency --no-graph --steps 8 --warmup 4"
bash tools/build_variant.sh prio3 -DADSP_SETPRIO=3 > $O/build1.log 2>&1 &
bash tools/build_variant.sh prio1 -DADSP_SETPRIO=1 > $O/build2.log 2>&1 &
wait
{
for r in 1 2; do for lib in "" abl/prio1.so abl/prio3.so; do
echo "[$lib] headline : $(ADSP_LIB=$lib $B 2>>$O/err.log | line)"
echo "[$lib] chain    : $(ADSP_LIB=$lib $B --no-stream-extra --filter chain --chunk 8192 --fs 96000 2>>$O/err.log | line)"
echo "[$lib] lc8192   : $(ADSP_LIB=$lib $B --no-stream-extra --chunk 8192 --channels 2048 2>>$O/err.log | line)"
echo "[$lib] eq512    : $(ADSP_LIB=$lib $B --no-stream-extra --filter eq3 --chunk 512 2>>$O/err.log | line)"
done; done
} 2>&1 | tee $O/shapes.txt
