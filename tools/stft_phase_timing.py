#!/usr/bin/env python
# NOTE (round 6): the timing / probe switches this tool builds with left the product sources; apply tools/ablation/lab_knobs_r06.patch
# (patch -p1 at the repo root) to a scratch tree first.
"""Debug: with a -DTAC_STFT_TIMING=1 build (TAC_AMD_LIB=...), print per-phase cycle sums per wave of the
complex-STFT / power-spectrogram kernel at cfg-2.   python tools/stft_phase_timing.py [stft|spec] [nblocks]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
what = sys.argv[1] if len(sys.argv) > 1 else 'stft'
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
layer = (tac.STFT(2048, 512) if what == 'stft' else tac.Spectrogram(2048, 512, power=2.)).cuda()
for _ in range(3):
    y = tac.realize(layer(x))
torch.cuda.synchronize()
# physical layout of the kernel output is (rows, T, F[,2]); the python view is a transpose of it
phys = (y.transpose(2, 3) if what == 'stft' else y.transpose(-2, -1)).contiguous().view(-1)
t = phys[:nblk * 4 * 16].view(nblk, 4, 16).cpu()[..., :12]
names = {0: 'loop top / store drain', 8: 'load + window', 2: 'pass0 butterflies', 3: 'pass1 readback+twiddle',
         4: 'pass1 butterflies', 5: 'pass2 readback+twiddle', 6: 'pass2 butterflies', 9: 'last-pass LDS write',
         1: 'next-frame load issue', 7: 'r2c split', 10: 'row staging (LDS)', 11: 'row store issue'}
tot = t.sum(-1)
frames_per_wave = 256 * 313 / (nblk * 4)
print('%s: per-wave total cycles: mean %.0f min %.0f max %.0f  (%.1f frames per wave -> %.0f cycles per frame)'
      % (what, tot.mean(), tot.min(), tot.max(), frames_per_wave, tot.mean() / frames_per_wave))
for i in (0, 8, 2, 3, 4, 5, 6, 9, 1, 7, 10, 11):
    col = t[..., i]
    print('%-26s %9.0f cycles/frame  (%.1f%%)' % (names[i], col.mean() / frames_per_wave, 100 * col.mean() / tot.mean()))
