#!/bin/bash
# round-6 GPU session 24: compressor / gate with time across the lanes of a wave (compressor_wave_kernel) against one lane per channel:
# bit-exactness (goldens, oracle, the two kernels against each other), the harness timing, throughput at 64 ... 16384 channels
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s24
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_recursive.py tests/test_gpu_moduletests.py tests/test_gpu_callers.py -q -m gpu -rf -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -15 | tee $O/pytest_subset.txt
timeout 600 python examples/harness_timing.py > $O/harness_timing.json 2> $O/harness_timing.err; echo "harness rc=$?"
python -c 'import json; d=json.load(open("gpurun_out/r6s24/harness_timing.json"))["ModuleTests.py"]; print({k: v["ms_per_chunk"] for k, v in d.items() if isinstance(v, dict)})'
python - <<'PY' | tee $O/throughput.txt
import os, time, json, torch, sys
sys.path.insert(0, os.getcwd())
import pyaudiodsptools_amd as adsp
print("# compressor / gate, device-resident float32 [steps, C, N], us per launch and Msamples/s: time across lanes (wave) vs one lane per channel (lane)")
for C, N, steps in [(1, 512, 1), (64, 512, 8), (1024, 512, 8), (4096, 512, 8), (4096, 4096, 8), (8192, 4096, 4), (16384, 4096, 2)]:
    adsp.config.initialize(44100, N)
    x = (torch.rand((steps, C, N), device="cuda") * 2 - 1) * 0.3
    y = torch.empty_like(x)
    row = {"C": C, "N": N, "steps": steps}
    for kind, mk in (("compressor", lambda: adsp.CreateCompressor(channels=C)), ("gate", lambda: adsp.CreateGate(channels=C))):
        for mode in ("wave", "lane"):
            if mode == "lane": os.environ["ADSP_SCAN_LANE_PER_CHANNEL"] = "1"
            else: os.environ.pop("ADSP_SCAN_LANE_PER_CHANNEL", None)
            eng = mk().engine
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < 0.2:
                eng.apply_device(x, y, steps); torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5): eng.apply_device(x, y, steps, torch.cuda.current_stream().cuda_stream)
            t1.record(); torch.cuda.synchronize()
            us = t0.elapsed_time(t1) / 5 * 1000
            row[f"{kind}_{mode}_us"] = round(us, 1); row[f"{kind}_{mode}_msps"] = round(steps * C * N / us, 1)
    os.environ.pop("ADSP_SCAN_LANE_PER_CHANNEL", None)
    print(json.dumps(row))
PY
