#!/bin/bash
# build a tuning variant of libadsp (the libadsp_tuning.so flavour: -DADSP_TUNING_BUILD, plans_var.hip linked in) into abl/<name>.so:
#   tools/build_variant.sh name -DADSP_MIN_WAVES=5 ...      run with ADSP_LIB=abl/<name>.so
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d)
mkdir -p "$root/abl"
cd "$root/pyaudiodsptools_amd/csrc"
for f in adsp_capi adsp_ring adsp_host adsp_effects adsp_rccl adsp_delay adsp_scan adsp_exact adsp_synth adsp_upols plans_f32 plans_s16 plans_s16_f64 plans_f32_epi plans_var plans_live; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-function -fno-slp-vectorize -DADSP_TUNING_BUILD "$@" -c -o $tmp/$f.o $f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/abl/$name.so" $tmp/*.o -ldl
rm -rf $tmp
echo "built abl/$name.so"
