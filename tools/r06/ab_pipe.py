#!/usr/bin/env python
"""Same-process A/B of the two forms of the fft_length-2048 kernels — transform on the VALU (tac_set_fft_pipe(0)) or on the
matrix pipe (1) — through the C ABI, alternating on the same device buffers.
    python tools/r06/ab_pipe.py mel|spec|stft [N]
cfg-2 (256 x 160 000 samples, 2048 / 512, 128 mel + dB), four rotating input batches, N launches per round (default 100),
nine rounds per form.  First checks the two forms against each other (the golden-vector test of the matrix-pipe form is
tests/test_gpu_parity.py::test_g2_melspectrogram_on_the_matrix_pipe)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd._native import StftDesc

op = sys.argv[1] if len(sys.argv) > 1 else 'mel'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
h = tac._native.lib()
P, I32, F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
dev = torch.device('cuda', 0)
rows, L, n_fft, hop, n_mels = 256, 160000, 2048, 512, 128
T = 1 + L // hop
torch.manual_seed(0)
xs = [torch.rand(rows, L, device=dev) * 2 - 1 for _ in range(4)]
window = torch.hann_window(n_fft, device=dev)
fb = tac.create_mel_filter(n_fft // 2 + 1, n_mels, 0.0, 8000.0, False).to(dev).contiguous()
stream = P(torch.cuda.current_stream().cuda_stream)
wpack = torch.empty(24576, device=dev); dsc = torch.empty(4096, dtype=torch.int32, device=dev); info = (I32 * 8)()
assert h.tac_melbank_pack(P(fb.data_ptr()), 1025, n_mels, n_fft, P(wpack.data_ptr()), 24576, P(dsc.data_ptr()), 4096, ctypes.cast(info, P), stream) == 0


def launch(x, out, db=1, power=2.0, r=rows, length=L):
    desc = StftDesc(r, length, length, n_fft, hop, n_fft, 1, 1, 0, 1, 0)
    if op == 'mel':
        rc = h.tac_melspec_sparse_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), power, P(wpack.data_ptr()), P(dsc.data_ptr()),
                                      ctypes.cast(info, P), n_mels, db, 1.0, 1e-7, P(out.data_ptr()), stream)
    elif op == 'spec':
        rc = h.tac_spectrogram_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), power, db, 1.0, 1e-7, P(out.data_ptr()), stream)
    else:
        rc = h.tac_stft_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), P(out.data_ptr()), stream)
    assert rc == 0, rc


width = {'mel': n_mels, 'spec': 1025, 'stft': 2050}[op]
outs = {}
for mode in (0, 1):
    h.tac_set_fft_pipe(mode)
    for db, power in ((0, 2.0), (1, 2.0), (0, 1.0)) if op != 'stft' else ((0, 2.0),):
        out = torch.full((rows * T * width,), float('nan'), device=dev)
        launch(xs[0], out, db, power)
        torch.cuda.synchronize()
        outs[(mode, db, power)] = out
    print('pipe %d route: %s' % (mode, h.tac_last_route().decode()))
for key in [k for k in outs if k[0] == 0]:
    a, b = outs[key], outs[(1,) + key[1:]]
    assert torch.isfinite(b).all(), 'non-finite values in the matrix-pipe form %r' % (key,)
    d = (a - b).abs()
    if key[1]:
        print('db=%d power=%g: max |valu - mfma| = %.3g dB (mean %.3g)' % (key[1], key[2], d.max().item(), d.mean().item()))
    else:
        print('db=%d power=%g: max |valu - mfma| = %.3g of max %.3g (%.3g relative to max)' % (key[1], key[2], d.max().item(), a.abs().max().item(), (d.max() / a.abs().max()).item()))

# timing
out = torch.empty(rows * T * width, device=dev)
res = {0: [], 1: []}
for rnd in range(9):
    for mode in (0, 1):
        h.tac_set_fft_pipe(mode)
        for i in range(10): launch(xs[i % 4], out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(N): launch(xs[i % 4], out)
        e1.record(); e1.synchronize()
        res[mode].append(e0.elapsed_time(e1) / N)
for mode in (0, 1):
    v = sorted(res[mode])
    print('pipe %d: median %.4f ms  (min %.4f, max %.4f)  %.1f M frames/s' % (mode, v[len(v) // 2], v[0], v[-1], rows * T / v[len(v) // 2] / 1e3))
d = sorted((b - a) / a for a, b in zip(res[0], res[1]))
print('mfma vs valu: median of per-round differences %+.1f %%' % (100 * d[len(d) // 2]))
h.tac_set_fft_pipe(-1)
