#!/usr/bin/env python
"""Host-side proof that csrc/exact_math.hpp reproduces torch-CPU float32 log1p (what the reference's
mu_law_encoding evaluates, functional.py:333) bit for bit, and that the closed-form encoder built on it
reproduces the reference codes for several n_quantize / input ranges.

    python tools/check_log1p_replica.py [--full]

Compiles the header with g++ (-ffp-contract=off: only the explicit fmaf's may fuse) into a scratch .so and
compares against torch on this host.  --full sweeps every float32 in [0, 256) (about 1.1e9 values, minutes);
the default strides through the same range.  Test infrastructure only — nothing here ships.
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = r'''
#include "exact_math.hpp"
extern "C" void run_log1p(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = tac::exact_log1pf(x[i]); }
extern "C" void run_enc(const float* x, long long* y, long n, int nq) {
    const float mu = (float)(nq - 1), l = tac::exact_log1pf(mu);
    for (long i = 0; i < n; ++i) {
        const float v = x[i];
        const float sgn = (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : v);
        const float comp = sgn * tac::exact_log1pf(mu * std::fabs(v)) / l;
        const float q = (comp + 1.0f) / 2.0f * mu + 0.5f;
        y[i] = !(std::fabs(q) < 9.2233720e18f) ? (long long)0x8000000000000000ULL : (long long)q;
    }
}
'''


def build():
    d = tempfile.mkdtemp(prefix='tac_log1p_')
    src = os.path.join(d, 'shim.cpp')
    open(src, 'w').write(SHIM)
    so = os.path.join(d, 'shim.so')
    subprocess.check_call(['g++', '-O2', '-mfma', '-ffp-contract=off', '-Wno-unknown-pragmas', '-shared', '-fPIC',
                           '-I', os.path.join(ROOT, 'torchaudio-contrib_amd', 'csrc'), src, '-o', so])
    return ctypes.CDLL(so)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def main():
    full = '--full' in sys.argv
    lib = build()
    bad = 0
    step = 1 if full else 61
    lo, hi = 0, 0x43800000                                   # bit patterns of [0, 256)
    chunk = 1 << 26
    for start in range(lo, hi, chunk * step):
        bits = np.arange(start, min(hi, start + chunk * step), step, dtype=np.int64).astype(np.uint32)
        x = bits.view(np.float32)
        y = np.empty_like(x)
        lib.run_log1p(ptr(x), ptr(y), ctypes.c_long(x.size))
        want = torch.log1p(torch.from_numpy(x)).numpy()
        bad += int((want.view(np.int32) != y.view(np.int32)).sum())
    print('log1p [0,256) stride %d: %d mismatches' % (step, bad))
    bits = np.arange(0x43800000, 0x7e000000, 997, dtype=np.int64).astype(np.uint32)
    x = bits.view(np.float32)
    y = np.empty_like(x)
    lib.run_log1p(ptr(x), ptr(y), ctypes.c_long(x.size))
    m = int((torch.log1p(torch.from_numpy(x)).numpy().view(np.int32) != y.view(np.int32)).sum())
    print('log1p [256, 4e37) stride 997: %d mismatches of %d' % (m, x.size))
    bad += m
    rng = np.random.default_rng(1)
    for nq in (256, 65536, 16, 2, 1024, 7):
        for scale in (1.0, 4.0, 1e3):
            x = ((rng.random(4_000_000, dtype=np.float32) * 2 - 1) * scale).astype(np.float32)
            y = np.empty(x.size, np.int64)
            lib.run_enc(ptr(x), ptr(y), ctypes.c_long(x.size), ctypes.c_int(nq))
            t = torch.from_numpy(x)
            mu = torch.tensor(nq - 1, dtype=t.dtype)
            want = ((t.sign() * torch.log1p(mu * t.abs()) / torch.log1p(mu) + 1) / 2 * mu + 0.5).long().numpy()
            m = int((want != y).sum())
            bad += m
            print('encode n_quantize=%d scale=%g: %d mismatches' % (nq, scale, m))
    print('OK' if bad == 0 else 'MISMATCH')
    return 0 if bad == 0 else 1


if __name__ == '__main__':
    sys.exit(main())
