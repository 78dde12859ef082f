#!/usr/bin/env python
"""Same-process A/B of the forms of the fft_length-2048 |X|^2 rows (stft_ring3_kernel): TAC_R3_FORM is read at every launch (lab
builds), the forms alternate on the same buffers.   python tools/r06/ab_ring_forms.py [form ...]   (default: all)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd._native import StftDesc

forms = sys.argv[1:] or ['base', 'h12', 'h12b', 'h15', 'h15b']
h = tac._native.lib()
P, F = ctypes.c_void_p, ctypes.c_float
dev = torch.device('cuda', 0)
rows, L, n_fft, hop = 256, 160000, 2048, 512
T = 1 + L // hop
torch.manual_seed(0)
xs = [torch.rand(rows, L, device=dev) * 2 - 1 for _ in range(4)]
window = torch.hann_window(n_fft, device=dev)
stream = P(torch.cuda.current_stream().cuda_stream)
desc = StftDesc(rows, L, L, n_fft, hop, n_fft, 1, 1, 0, 1, 0)


def launch(form, x, out, power=2.0, db=0):
    if form == 'base': os.environ.pop('TAC_R3_FORM', None)
    else: os.environ['TAC_R3_FORM'] = form
    rc = h.tac_spectrogram_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), power, db, 1.0, 1e-7, P(out.data_ptr()), stream)
    assert rc == 0, (form, rc)


ref = {}
for form in forms:
    for power, db in ((2.0, 0), (1.0, 0), (2.0, 1)):
        out = torch.full((rows * T * 1025,), float('nan'), device=dev)
        launch(form, xs[0], out, power, db)
        torch.cuda.synchronize()
        if form == forms[0]:
            ref[(power, db)] = out
        else:
            same = torch.equal(out, ref[(power, db)])
            d = (out - ref[(power, db)]).abs().max().item()
            print('%-5s power %g db %d: %s (max |diff| %.3g)  route %s' % (form, power, db, 'bit-identical' if same else 'DIFFERENT', d, h.tac_last_route().decode()))
out = torch.empty(rows * T * 1025, device=dev)
N = 100
res = {f: [] for f in forms}
for rnd in range(9):
    order = forms if rnd % 2 == 0 else forms[::-1]
    for form in order:
        for i in range(10): launch(form, xs[i % 4], out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(N): launch(form, xs[i % 4], out)
        e1.record(); e1.synchronize()
        res[form].append(e0.elapsed_time(e1) / N)
base = sorted(res[forms[0]])[4]
for form in forms:
    v = sorted(res[form])
    d = sorted((b - a) / a for a, b in zip(res[forms[0]], res[form]))
    print('%-5s median %.4f ms (min %.4f max %.4f) = %.1f %% of 8 TB/s | vs %s: %+.1f %%' % (form, v[4], v[0], v[-1], 100 * 6148.0 * rows * T / (v[4] * 1e-3) / 8e12, forms[0], 100 * d[4]))
os.environ.pop('TAC_R3_FORM', None)
