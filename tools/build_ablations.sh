#!/bin/bash
# Ablation builds for the speed-of-light model of the batch kernels (profiles/r5_sol_model.md): libadsp with ONE translation unit
# (plans_f32.hip: the float32 plain kernels bench.py's batch mode runs) recompiled with -DADSP_ABLATE=<mask>, the other objects taken
# from the in-tree build.  Results of such a library are WRONG by construction (an ingredient of the kernel is missing); they are
# timed with bench.py --no-parity-check only.   usage: tools/build_ablations.sh [mask ...]   -> abl/abl<mask>.so
# masks (fftconv_kernel.hpp): 1 no pass-twiddle loads, 2 no pair-table loads, 4 no LDS exchange, 8 no global input loads,
# 16 no output stores, 32 butterflies replaced by copies, 64 I/O aliased onto 8 channels (L2-resident), 256 no spectrum stage
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/pyaudiodsptools_amd/csrc"
make -j8 tuning >/dev/null   # the tuning flavour's objects (tuning/*.o): ablation switches do not compile into the product library
mkdir -p "$root/abl"
masks=${@:-"0 8 16 24 4 28 32 60 3 64 256"}
# "persist" instead of a mask: -DADSP_PERSIST=1 (a workgroup loops over KernelArgs::blk_iters consecutive blocks; results stay correct;
# run with ADSP_PERSIST_BUILD=1 ADSP_BLK_ITERS=<n> ADSP_LIB=abl/persist.so)
others=$(ls tuning/*.o | grep -v '/plans_f32\.o$')
tmp=$(mktemp -d)
n=0
for m in $masks; do
  if [ "$m" = persist ]; then
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-function -fno-slp-vectorize -DADSP_TUNING_BUILD -DADSP_PERSIST=1 -c -o $tmp/plans_f32_p.o plans_f32.hip \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/abl/persist.so" $tmp/plans_f32_p.o $others -ldl && echo "built abl/persist.so" ) &
    continue
  fi
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-function -fno-slp-vectorize -DADSP_TUNING_BUILD -DADSP_ABLATE=$m -c -o $tmp/plans_f32_$m.o plans_f32.hip \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/abl/abl$m.so" $tmp/plans_f32_$m.o $others -ldl && echo "built abl/abl$m.so" ) &
  n=$((n+1)); [ $((n % 6)) -eq 0 ] && wait
done
wait
rm -rf $tmp
