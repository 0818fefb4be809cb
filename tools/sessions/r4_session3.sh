#!/bin/bash
# round-4 GPU session 3: debugging the live sessions (host publication not seen by the relay in some sessions)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s3
ADSP_DEBUG=1 timeout 300 python - > gpurun_out/r4s3/debug.txt 2>&1 <<'PY'
import ctypes, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pyaudiodsptools_amd as adsp
from pyaudiodsptools_amd import FirEngine, FirStream, design
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
fs = 44100
def one(n, kind, channels, delay=0.0, keep=None):
    taps = design.lowcut_kernel(500, fs, n) if kind == "lowcut" else design.eq3_composite(100, 2, 700, -4, 8000, 5, fs, n)
    fir = FirStream(taps, n)
    eng = FirEngine(fir, channels=channels, ring_slots=7)
    x = torch.empty((6, channels, n), device="cuda").uniform_(-1, 1)
    out = torch.full((3, channels, n), float("nan"), device="cuda")
    cons = torch.cuda.Stream()
    eng.live_configure(step_timeout_ms=100.0)
    eng.live_start(out, 3, 4, None)
    if delay:
        time.sleep(delay)
    ok = True
    for k in range(4):
        slot = eng.live_slot()
        hip.hipMemcpyAsync(slot, x[k].data_ptr(), channels * n * 4, 3, None)
        torch.cuda.current_stream().synchronize()
        eng.live_publish()
        try:
            eng.live_wait(k + 1, 2000.0)
        except Exception as exc:
            ok = False
            break
    try:
        eng.live_stop()
    except Exception as exc:
        pass
    print("==", n, kind, channels, "delay", delay, "OK" if ok else "FAILED", flush=True)
    sys.stderr.flush()
    if keep is not None:
        keep.append(eng)
    else:
        eng.close()
print("--- same config five times")
for i in range(5):
    one(512, "eq", 33)
print("--- with a 5 ms pause between start and publish")
for i in range(3):
    one(512, "eq", 33, delay=0.005)
print("--- engines kept alive (no host free between sessions)")
keep = []
for i in range(6):
    one(512, "eq", 33, keep=keep)
for cfg in [(1024, "eq", 33), (512, "lowcut", 70), (256, "eq", 5), (512, "eq", 33), (512, "eq", 34), (512, "eq", 66)]:
    one(*cfg, keep=keep)
PY
grep -E "^==|^---|relay" gpurun_out/r4s3/debug.txt | sed 's/progress:.*relay/relay/; s/| [0-9 ]*$//' | cut -c1-330
