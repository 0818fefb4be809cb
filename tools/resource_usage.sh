#!/bin/bash
# print per-kernel register/occupancy summary for the HIP library (no GPU needed), with the flags of csrc/Makefile
cd "$(dirname "$0")/../pyaudiodsptools_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -fno-slp-vectorize -shared -Rpass-analysis=kernel-resource-usage $EXTRA -o /tmp/_adsp_ru.so plans_f32.hip 2>&1 \
 | grep -E "Function Name|Name:|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs:" \
 | sed -E 's/.*remark: [^ ]+ +//; s/\[-Rpass.*//; s/.*fftconv_kernelINS_4PlanI/Plan /; s/EEELi([0-9]+)EEEvNS.*/ CPB=\1/' | paste -sd' ' | sed 's/Plan /\nPlan /g'
echo
