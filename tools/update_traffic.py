#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes of a tools/profile_gpu.sh run into profiles/traffic.json, stamped with the
identity of the kernel sources they were measured on (bench.py reports `traffic: null` when the stamp is stale).

usage: tools/update_traffic.py <key> <gpurun_out/prof_TAG/summary.txt> <steps_per_launch> <algorithmic_bytes_per_launch> <profiles/NAME_summary.txt> [note]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sha16  # noqa: E402

key, summary, cps, alg, source = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
txt = open(summary).read()
fetch = float(re.search(r"FETCH_SIZE\s+per-dispatch avg\s+([0-9.]+)", txt).group(1))
write = float(re.search(r"WRITE_SIZE\s+per-dispatch avg\s+([0-9.]+)", txt).group(1))
path = os.path.join(ROOT, "profiles", "traffic.json")
d = json.load(open(path))
d[key] = {"steps_per_launch": cps, "FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1),
          "hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "algorithmic_bytes_per_launch": alg,
          "source": source, "kernel_sha16": kernel_sha16()}
if len(sys.argv) > 6:
    d[key]["note"] = sys.argv[6]
json.dump(d, open(path, "w"), indent=1)
print(key, d[key], "ratio", round(d[key]["hbm_bytes_per_launch"] / alg, 3))
