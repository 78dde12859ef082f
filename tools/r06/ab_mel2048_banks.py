#!/usr/bin/env python
"""Same-process A/B of library builds on the fused fft_length-2048 chain (cfg-2: 256 x 160 000 samples, hop 512, power 2, dB) with a bank of
n_mels bands: every library packs the bank itself.    python tools/r06/ab_mel2048_banks.py n_mels:sample_rate name=path [name=path ...]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd._native import StftDesc
P, F, I = ctypes.c_void_p, ctypes.c_float, ctypes.c_int32
args = sys.argv[1:]
n_mels, sr = (int(v) for v in args.pop(0).split(':'))
dev = torch.device('cuda', 0)
fb = tac.MelFilterbank(num_mels=n_mels, sample_rate=sr, num_freqs=1025).get_filterbank().to(dev).contiguous()
rows, L, n_fft, hop = 256, 160000, 2048, 512
T = 1 + L // hop
xs = [torch.rand(rows, L, device=dev) * 2 - 1 for _ in range(4)]
window = torch.hann_window(n_fft, device=dev)
desc = StftDesc(rows, L, L, n_fft, hop, n_fft, 1, 1, 0, 1, 0)
out = torch.empty(rows * T * n_mels, device=dev)
stream = P(torch.cuda.current_stream().cuda_stream)
libs = []
for a in args:
    name, path = a.split('=', 1)
    h = ctypes.CDLL(os.path.abspath(path))
    h.tac_melbank_pack.argtypes = [P, I, I, I, P, I, P, I, P, P]
    h.tac_melspec_sparse_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), F, P, P, P, I, ctypes.c_int, F, F, P, P]
    wpack = torch.empty(24576, dtype=torch.float32, device=dev)
    dsc = torch.empty(8192, dtype=torch.int32, device=dev)
    info = (I * 8)()
    rc = h.tac_melbank_pack(P(fb.data_ptr()), 1025, n_mels, 2048, P(wpack.data_ptr()), 24576, P(dsc.data_ptr()), 8192, ctypes.cast(info, P), stream)
    print('%-10s pack rc %d info %s' % (name, rc, [int(v) for v in info]))
    if rc == 0: libs.append((name, h, wpack, dsc, info))
def launch(e, x):
    name, h, wpack, dsc, info = e
    rc = h.tac_melspec_sparse_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), 2.0, P(wpack.data_ptr()), P(dsc.data_ptr()),
                                  ctypes.cast(info, P), n_mels, 1, 1.0, 1e-7, P(out.data_ptr()), stream)
    assert rc == 0, (name, rc)
ref = None
for e in libs:
    launch(e, xs[0]); torch.cuda.synchronize()
    cur = out.clone()
    if ref is None: ref = cur
    else: print('check %-10s max |diff| vs %s: %.3g dB' % (e[0], libs[0][0], (cur - ref).abs().max().item()))
for _ in range(50):
    for e in libs: launch(e, xs[0])
torch.cuda.synchronize()
ev = {e[0]: [] for e in libs}; k = 0
for r in range(300):
    for e in (libs if r % 2 == 0 else libs[::-1]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(e, xs[k % 4]); b.record(); k += 1
        ev[e[0]].append((a, b))
torch.cuda.synchronize()
ts = {n: [a.elapsed_time(b) for a, b in v] for n, v in ev.items()}
q = lambda v, f: sorted(v)[int(f * (len(v) - 1))]
base = libs[0][0]
for e in libs:
    name = e[0]
    d = sorted(x - y for x, y in zip(ts[name], ts[base]))
    print('mel2048 %d bands %-8s median %.4f ms  p10 %.4f  p90 %.4f | vs %s %+.2f %%' % (n_mels, name, q(ts[name], .5), q(ts[name], .1), q(ts[name], .9), base, 100 * q(d, .5) / q(ts[base], .5)))
