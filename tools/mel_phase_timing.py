#!/usr/bin/env python
"""Debug: with a TAC_MEL_TIMING=1 build (TAC_AMD_LIB=...), print per-phase cycle sums per wave."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
m = tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512).cuda()
for _ in range(3):
    y = m(x)
torch.cuda.synchronize()
phys = y.transpose(-2, -1).contiguous().view(-1)
t = phys[:256 * 8 * 8].view(256, 8, 8).cpu()
names = ['tile-start', 'phase A', 'barrier A', 'phase B', 'barrier B', 'phase C']
tot = t[..., :6].sum(-1)
print('per-wave total cycles: mean %.0f  min %.0f  max %.0f' % (tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names):
    col = t[..., i]
    print('%-10s mean %9.0f (%.1f%%)  per-wave-index means: %s' % (n, col.mean(), 100 * col.mean() / tot.mean(),
          ' '.join('%7.0f' % v for v in col.mean(0))))
