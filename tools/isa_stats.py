#!/usr/bin/env python3
"""Compile adsp_capi.hip with -save-temps and print an instruction histogram per kernel (no GPU needed).
usage: tools/isa_stats.py [substring-of-mangled-name] [extra hipcc flags...]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "Li4096ELi32"
extra = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="isa_")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-save-temps",
       "-o", os.path.join(tmp, "x.so"), os.path.join(ROOT, "pyaudiodsptools_amd/csrc/plans_f32.hip")] + extra
subprocess.run(cmd, cwd=tmp, check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "plans_f32-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for f in re.split(r'\n\s*\.globl\s+', s):
    name = f.split('\n', 1)[0].strip()
    if pat not in name or 'fftconv' not in name:
        continue
    ops = collections.Counter()
    for line in f.split('\n'):
        line = line.strip()
        m = re.match(r'^([a-z_0-9]+)(\s|$)', line)
        if m and not line.startswith(('.', ';')):
            ops[m.group(1)] += 1
    g = collections.Counter()
    for k, v in ops.items():
        key = ('v_pk' if k.startswith('v_pk_') else 'v_mov' if k.startswith('v_mov') or k.startswith('v_accvgpr') else 'valu' if k.startswith('v_') else
               'waitcnt' if k.startswith('s_waitcnt') else 'barrier' if k == 's_barrier' else 'salu' if k.startswith('s_') else
               'lds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'scratch_')) else 'other')
        g[key] += v
    vg = re.search(r'\.vgpr_count:\s+(\d+)', f) or re.search(r'; NumVgprs: (\d+)', f)
    print(name[:90], 'total', sum(ops.values()), dict(g), 'vgpr', vg.group(1) if vg else '?')
    if os.environ.get("TOP"):
        for k, v in ops.most_common(int(os.environ["TOP"])):
            print('   ', k, v)
