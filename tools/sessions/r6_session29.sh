#!/bin/bash
# round-6 GPU session 29: the long-kernel engine with the real-FFT split taken ONCE per block (forward launch) and the re-packing once per output
# block (multiply launch): the delay line keeps split spectra, the multiply-accumulate is one complex multiply-add per register and partition.
# Parity (the long-kernel tests), then alternating A/B against the library before the change (abl/presplit.so).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s29
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_moduletests.py tests/test_gpu_fuzz.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15 | tee $O/pytest_subset.txt
for r in 1 2; do for l in presplit default; do
  if [ "$l" = default ]; then lib=""; else lib="abl/$l.so"; fi
  echo "== lib=[$l]" | tee -a $O/ab.txt
  ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols 2>/dev/null | tail -1 | tee -a $O/ab.txt | cut -c1-1200
  for b in 8192 16384; do echo "-- block $b" | tee -a $O/ab.txt; ADSP_LIB=$lib timeout 600 python tools/bench_upols.py --only upols --block $b --channels 1024 2>/dev/null | tail -1 | tee -a $O/ab.txt | cut -c1-700; done
done; done
