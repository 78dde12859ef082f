#!/usr/bin/env python
"""Compare the gfx950 assembly of every kernel of two builds, instruction by instruction (comments, directives and the
function numbers inside basic-block labels stripped): used in round 5 to check that taking the A/B switches out of the
product headers changed no kernel.
    for f in melspec_sparse stft_kernels backward stft_small stft_n400; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -save-temps=obj -c csrc/$f.hip -o DIR/$f.o; done
    python tools/r05/compare_isa.py DIR_BEFORE DIR_AFTER"""
import glob, hashlib, os, re, sys


def kernels(path):
    s = open(path).read()
    out = {}
    for part in re.split(r'\n(?=_Z[\w]+:\s+; @)', s)[1:]:
        name = part.split(':', 1)[0]
        body = part.split('\n.Lfunc_end', 1)[0]
        lines = [re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r';.*$', '', l)).rstrip() for l in body.splitlines()]
        lines = [l for l in lines if l.strip() and not (l.strip().startswith('.') and not l.strip().startswith('.LBB_'))]
        out[name] = hashlib.sha1('\n'.join(lines).encode()).hexdigest()
    return out


before, after = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(os.path.join(before, '*-hip-amdgcn-amd-amdhsa-gfx950.s'))):
    g = os.path.join(after, os.path.basename(f))
    a, b = kernels(f), kernels(g)
    diff = [k for k in a if k in b and a[k] != b[k]]
    print('%-16s kernels %3d  identical %3d  different %d  gone %d  new %d' % (
        os.path.basename(f).split('-hip')[0], len(a), sum(1 for k in a if k in b and a[k] == b[k]), len(diff),
        sum(1 for k in a if k not in b), sum(1 for k in b if k not in a)))
    for k in diff:
        print('   DIFF', k[:150])
