#!/usr/bin/env python
# NOTE (round 6): the timing / probe switches this tool builds with left the product sources; apply tools/ablation/lab_knobs_r06.patch
# (patch -p1 at the repo root) to a scratch tree first.
"""Debug: with a -DTAC_ST_TIMING=1 build (TAC_AMD_LIB=...), per-wave cycle sums of the streaming mel kernel at cfg-2 by
stage (either thread of the wave): s0 window + pass 0, s1 pass 1, s2 pass 2, s3 R2C + |X|^2 row (+ next request),
s4 contraction + dB + store."""
import os, sys
os.environ.setdefault('TAC_STREAM2', '1')      # the stamps live in round 2's two-frame rotation kernel
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                        tac.AmplitudeToDb()).cuda()
xs = [x] + [torch.rand(256, 1, 160000, device='cuda') * 2 - 1 for _ in range(int(os.environ.get('TAC_ROTATE', '1')) - 1)]
for i in range(3 * len(xs) + 1):            # (TAC_ROTATE=4: the stamped launch reads a batch that left the Infinity Cache)
    y = m(xs[i % len(xs)])
torch.cuda.synchronize()
phys = y.transpose(-2, -1).contiguous().view(-1)
W = int(os.environ.get('TAC_ST_WAVES', '8'))
NS = 12
t = phys[:256 * 8 * NS].view(256, 8, NS).cpu()[:, :W]
frames = 256 * 313 / (256 * W)
tot = t[..., 0].mean()
print('per wave: %.0f cycles total, %.1f frames -> %.0f cycles per frame (two frames in flight)' % (tot, frames, tot / frames))
print('total by wave: ' + ' '.join('%7.0f' % v for v in t[..., 0].mean(0)))
for k, name in ((8, '  s0 fine: previous drain + butterflies'), (9, '  s0 fine: exchange write issue'),
                (10, '  s0 fine: write drain + read issue'), (3, 'wait for the samples (vmcnt)'), (1, 's0 window + pass 0 + exchange'), (2, 's12 pass 1, in-register exchange, pass 2'), (4, 's3 r2c + row + request'),
                (5, 's4 contraction + dB + store')):
    col = t[..., k]
    print('%-40s %7.0f cycles/frame (%4.1f%%)  by wave: %s' % (name, col.mean() / frames, 100 * col.mean() / tot,
                                                              ' '.join('%5.0f' % (v / frames) for v in col.mean(0))))
