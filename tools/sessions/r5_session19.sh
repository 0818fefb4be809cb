#!/bin/bash
# round-5 GPU session 19: the multiply launch of the long-kernel engines on 16-byte table entries per pair and partition ((2s, 2d); the
# matrix entries formed in the kernel from them and the bin's twiddle) against 24-byte entries (c1, c2, c4: build_ab/libadsp_head.so),
# alternating on one box, both block sizes; the long-kernel tests (parity against the float64 direct sum) on the new tables.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r5s19
mkdir -p $O
for r in 1 2; do
  for lib in "" build_ab/libadsp_head.so; do
    for b in 8192 16384; do
      echo "== lib=[${lib:-product}] block $b" | tee -a $O/upols_ab.txt
      if [ -z "$lib" ]; then timeout 300 python tools/bench_upols.py --only upols --block $b 2>&1 | tail -1 | tee -a $O/upols_ab.txt
      else ADSP_LIB=$PWD/$lib timeout 300 python tools/bench_upols.py --only upols --block $b 2>&1 | tail -1 | tee -a $O/upols_ab.txt; fi
    done
  done
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "upols or example4 or long_kernel or partition or smoke" 2>&1 | tail -5 | tee $O/tests.txt
python - <<'PY' | tee $O/parity.txt
import sys, torch
sys.path.insert(0, ".")
import bench
f = bench.long_kernel_figures(torch.device("cuda", 0), channels=16, calls=4)
print({k: (v["us_per_call"], v["max_rel_err_vs_float64_direct_sum"]) for k, v in f.items() if isinstance(v, dict)})
PY
