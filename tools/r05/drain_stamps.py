#!/usr/bin/env python
"""Where the waves of stft_drain3i_kernel spend their cycles (a -DTAC_S3_DRAIN_STAMPS=1 build: per-wave cycle sums overwrite the
head of the output; WRONG RESULTS).   python tools/r05/drain_stamps.py <lib.so> <stft|spec> TW DW"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from torchaudio_contrib_amd._native import StftDesc
lib, op, TW, DW = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
P, F = ctypes.c_void_p, ctypes.c_float
h = ctypes.CDLL(os.path.abspath(lib))
h.tac_stft_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), P, P]
h.tac_spectrogram_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), F, ctypes.c_int, F, F, P, P]
dev = torch.device('cuda', 0)
rows, L, n_fft, hop = 256, 160000, 2048, 512
T = 1 + L // hop
xs = [torch.rand(rows, L, device=dev) * 2 - 1 for _ in range(4)]
window = torch.hann_window(n_fft, device=dev)
desc = StftDesc(rows, L, L, n_fft, hop, n_fft, 1, 1, 0, 1, 0)
stream = P(torch.cuda.current_stream().cuda_stream)
out = torch.empty(rows * T * (2050 if op == 'stft' else 1025), device=dev)
def launch(x):
    if op == 'stft': rc = h.tac_stft_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), P(out.data_ptr()), stream)
    else: rc = h.tac_spectrogram_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), 2.0, 0, 1.0, 1e-7, P(out.data_ptr()), stream)
    assert rc == 0
for i in range(20): launch(xs[i % 4])
torch.cuda.synchronize()
evs = []
for i in range(100):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); launch(xs[i % 4]); b.record(); evs.append((a, b))
torch.cuda.synchronize()
kms = sorted(a.elapsed_time(b) for a, b in evs)[50]
W = TW + DW
st = out[:256 * W * 8].view(256, W, 8).double().cpu()
prod, drn = st[:, :TW].mean(dim=(0, 1)), st[:, TW:].mean(dim=(0, 1))
tot = prod[7].item()
print('kernel %.4f ms (median of 100, events); a transform wave\'s frame loop is %.0f cycles -> >= %.0f MHz shader clock while it runs' % (kms, tot, tot / kms / 1e3))
print('%s %s TW=%d DW=%d: cycles per wave (mean over %d workgroups)' % (os.path.basename(lib), op, TW, DW, 256))
ring = len(sys.argv) > 5 and sys.argv[5] == 'ring'
names_p = ['wait samples', 'wait area', 'wait tail buffer', 'row writes + publish', 'window + butterfly 0', 'exchange + passes + partners', '-', 'TOTAL']
if ring: names_p = ['wait samples', 'wait slot', 'R2C split', 'row writes + publish (+ late request)', 'window + butterfly 0', 'exchange + passes + partners', 'request issue (EARLY)', 'TOTAL']
names_d = ['wait rows', 'span reads', '-', 'store issue', '-', '-', '-', 'TOTAL']
for n, v in zip(names_p, prod.tolist()):
    if n != '-': print('  transform  %-30s %10.0f  %5.1f %%' % (n, v, 100 * v / tot))
for n, v in zip(names_d, drn.tolist()):
    if n != '-': print('  drain      %-30s %10.0f  %5.1f %%' % (n, v, 100 * v / drn[7].item()))
