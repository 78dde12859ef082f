#!/usr/bin/env python
# NOTE (round 6): the timing / probe switches this tool builds with left the product sources; apply tools/ablation/lab_knobs_r06.patch
# (patch -p1 at the repo root) to a scratch tree first.
"""Debug: with a -DTAC_SP_TIMING=1 build (TAC_AMD_LIB=...), print per-phase cycle sums per wave of the fused
band-sparse mel kernel at cfg-2.   python tools/mel_phase_timing.py [nblocks]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                        tac.AmplitudeToDb()).cuda()
for _ in range(3):
    y = tac.realize(m(x))
torch.cuda.synchronize()
phys = y.transpose(-2, -1).contiguous().view(-1)
t = phys[:nblk * 8 * 16].view(nblk, 8, 16).cpu()[..., :12]
names = {0: 'tile top / phase C', 8: 'A: load + window', 2: 'A: pass0 butterflies', 3: 'A: pass1 readback+twiddle',
         4: 'A: pass1 butterflies', 5: 'A: pass2 readback+twiddle', 6: 'A: pass2 butterflies', 9: 'A: last-pass LDS write',
         10: 'A: r2c + power rows', 1: 'barrier A', 7: 'phase B (contraction)', 11: 'barrier B'}
tot = t.sum(-1)
frames_per_wave = 256 * 313 / (nblk * 8)
print('per-wave total cycles: mean %.0f min %.0f max %.0f  (%.1f frames per wave -> %.0f cycles per frame)'
      % (tot.mean(), tot.min(), tot.max(), frames_per_wave, tot.mean() / frames_per_wave))
for i in (8, 2, 3, 4, 5, 6, 9, 10, 1, 7, 11, 0):
    col = t[..., i]
    print('%-28s %9.0f cycles/frame  (%.1f%%)   by wave: %s' % (names[i], col.mean() / frames_per_wave,
          100 * col.mean() / tot.mean(), ' '.join('%6.0f' % (v / frames_per_wave) for v in col.mean(0))))
