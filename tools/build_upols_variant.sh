#!/bin/bash
# a variant of the long-kernel engine only: adsp_upols.hip recompiled with the given flags, linked against the product's other objects -> abl/<name>.so
#   tools/build_upols_variant.sh name -DADSP_UPOLS_MAC_WAVES=3 ...      run with ADSP_LIB=abl/<name>.so
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/abl"
cd "$root/pyaudiodsptools_amd/csrc"
make -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-function -fno-slp-vectorize "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/upols_$name.o adsp_upols.hip 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/\[-Rpass.*//; s/\[bytes\/lane\]//' | paste -sd' ' | sed 's/Function Name/\nFN/g' | grep "upols.*Lb0EEEvNS_9UpolsArgsE" | sed -E 's/: _ZN4adsp[0-9]+//; s/INS_4PlanI/ /; s/ELb0ELb1E[^ ]*Lb([01])EEEvNS_9UpolsArgsE/ /'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/abl/$name.so" $(ls *.o | grep -v adsp_upols.o) /tmp/upols_$name.o -ldl
echo "built abl/$name.so"
