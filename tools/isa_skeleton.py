#!/usr/bin/env python3
"""Order of memory operations, waits and barriers in the headline kernel's ISA (no GPU needed).
usage: tools/isa_skeleton.py [substring-of-mangled-name] [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "Li4096ELi16ELi3ELi16ELi16ELi16ELi1ELb1EEELi1ELi2ELb0ELb0"
tmp = tempfile.mkdtemp(prefix="isa_")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-slp-vectorize", "-save-temps", "-c",
                "-o", os.path.join(tmp, "x.o"), os.path.join(ROOT, "pyaudiodsptools_amd/csrc/plans_f32.hip")] + sys.argv[2:],
               cwd=tmp, check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "plans_f32-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def cls(l):
    if l.startswith('global_load'): return 'GL'
    if l.startswith('global_store'): return 'GS'
    if l.startswith(('ds_write', 'ds_store')): return 'DW'
    if l.startswith(('ds_read', 'ds_load')): return 'DR'
    if l.startswith('s_barrier'): return 'BAR'
    if l.startswith('s_waitcnt'): return 'W(' + l.split(None, 1)[1] + ')'
    if l.startswith('v_'): return 'v'
    if l.startswith(('s_cbranch', 's_branch')): return 'BR'
    return None


for f in re.split(r'\n\s*\.globl\s+', s):
    name = f.split('\n', 1)[0].strip()
    if pat in name and 'fftconv_kernel' in name:
        out, cnt, last = [], 0, None
        for l in (x.strip() for x in f.split('\n')):
            c = cls(l)
            if c is None:
                continue
            if c == last and c in ('v', 'GL', 'GS', 'DW', 'DR'):
                cnt += 1
            else:
                if last:
                    out.append(f"{last}x{cnt}" if cnt > 1 else last)
                last, cnt = c, 1
        out.append(f"{last}x{cnt}")
        txt = ' '.join(out)
        i = txt.find('DW')
        print(txt[i - 200:i + 2600])
        break
