#!/usr/bin/env python
"""Same-process A/B of library builds on the cfg-4 slice (64 x 480 000 samples, fft_length 4096 / hop 1024, |X| rows) through the C ABI.
    python tools/r05/ab_n4096.py name=path [name=path ...]"""
import ctypes, os, sys, time
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
rows, L, n_fft, hop = 64, 480000, 4096, 1024
T = 1 + L // hop
xs = [torch.rand(rows, L, device=dev) * 2 - 1 for _ in range(4)]
window = torch.hann_window(n_fft, device=dev)
desc = StftDesc(rows, L, L, n_fft, hop, n_fft, 1, 1, 0, 1, 0)
out = torch.empty(rows * T * (n_fft // 2 + 1), device=dev)
stream = P(torch.cuda.current_stream().cuda_stream)
def launch(h, x):
    assert h.tac_spectrogram_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), 1.0, 0, 1.0, 1e-7, P(out.data_ptr()), stream) == 0
ref = None
for name, h in libs:
    launch(h, xs[0]); torch.cuda.synchronize()
    cur = out.clone()
    if ref is None: ref = cur
    else: print('check %-10s max |diff| vs %s: %.3g' % (name, libs[0][0], (cur - ref).abs().max().item()))
for _ in range(50):
    for name, h in libs: launch(h, xs[0])
torch.cuda.synchronize()
ev = {n: [] for n, _ in libs}; k = 0
for r in range(300):
    for name, h in (libs if r % 2 == 0 else libs[::-1]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(h, xs[k % 4]); b.record(); k += 1
        ev[name].append((a, b))
torch.cuda.synchronize()
ts = {n: [a.elapsed_time(b) for a, b in v] for n, v in ev.items()}
q = lambda v, f: sorted(v)[int(f * (len(v) - 1))]
base = libs[0][0]
for name, _ in libs:
    d = sorted(x - y for x, y in zip(ts[name], ts[base]))
    print('n4096 %-10s median %.4f ms  p10 %.4f  p90 %.4f | vs %s %+.2f %%' % (name, q(ts[name], .5), q(ts[name], .1), q(ts[name], .9), base, 100 * q(d, .5) / q(ts[base], .5)))
