#!/usr/bin/env python
"""Same-process A/B of library builds on BASELINE configs[3] at FULL size (64 x 8ch x 2 880 000 samples, fft_length 4096 / hop 1024,
|X| rows: 5.9 GB in, 11.8 GB out, one launch) through the C ABI.    python tools/r06/ab_cfg4_full.py name=path [name=path ...]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from torchaudio_contrib_amd._native import StftDesc
P, F = ctypes.c_void_p, ctypes.c_float
libs = []
for a in sys.argv[1:]:
    name, path = a.split('=', 1)
    h = ctypes.CDLL(os.path.abspath(path))
    h.tac_spectrogram_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), F, ctypes.c_int, F, F, P, P]
    libs.append((name, h))
dev = torch.device('cuda', 0)
rows, L, n_fft, hop = 512, 2880000, 4096, 1024
T = 1 + L // hop
x = torch.rand(rows, L, device=dev) * 2 - 1
window = torch.hann_window(n_fft, device=dev)
desc = StftDesc(rows, L, L, n_fft, hop, n_fft, 1, 1, 0, 1, 0)
out = torch.empty(rows * T * (n_fft // 2 + 1), device=dev)
stream = P(torch.cuda.current_stream().cuda_stream)
def launch(h):
    assert h.tac_spectrogram_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), 1.0, 0, 1.0, 1e-7, P(out.data_ptr()), stream) == 0
ref = None
for name, h in libs:
    launch(h); torch.cuda.synchronize()
    cur = out[:50000000].clone()
    if ref is None: ref = cur
    else: print('check %-10s max |diff| vs %s (first 5e7 values): %.3g' % (name, libs[0][0], (cur - ref).abs().max().item()))
ts = {n: [] for n, _ in libs}
for r in range(12):
    for name, h in (libs if r % 2 == 0 else libs[::-1]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(h); b.record(); torch.cuda.synchronize()
        ts[name].append(a.elapsed_time(b))
alg = rows * T * (4 * hop + 4 * (n_fft // 2 + 1))
for name, _ in libs:
    v = sorted(ts[name][2:]); med = v[len(v) // 2]
    print('cfg4 %-10s median %.3f ms  min %.3f  max %.3f | %.0f GB/s = %.1f %% of 8 TB/s' % (name, med, v[0], v[-1], alg / med / 1e6, alg / med / 1e6 / 80))
