#!/usr/bin/env python
"""With a -DTAC_S3_STAMPS=1 build (TAC_AMD_LIB=...): cycles per frame and wave of melspec_stream3_kernel's stages at cfg-2
(s_memtime stamps; every stamp drains lgkmcnt, so the stages add up to more than the unstamped loop).  TAC_ROTATE batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                        tac.AmplitudeToDb()).cuda()
nrot = int(os.environ.get('TAC_ROTATE', '4'))
xs = [torch.rand(256, 1, 160000, device='cuda') * 2 - 1 for _ in range(nrot)]
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 0.6:
    for _ in range(10):
        y = m(xs[k % nrot]); k += 1
    torch.cuda.synchronize()
acc = torch.zeros(8, dtype=torch.float64)
for r in range(20):
    y = m(xs[k % nrot]); k += 1
    torch.cuda.synchronize()
    acc += y.transpose(-2, -1).contiguous().view(-1)[:256 * 12 * 8].view(256 * 12, 8).double().mean(0).cpu()
acc /= 20
frames_per_wave = 313 / 12.0
names = ['s0a wait+window+butterfly0', 's0b exchange write+readback', 's12 passes 1,2 + half write', 's3 partners+r2c+row', 'request', 's4 contraction+dB+store']
tot = float(acc[:6].sum())
for n, v in zip(names, acc[:6].tolist()):
    print('%-32s %8.0f cycles/frame  %5.1f %%' % (n, v / frames_per_wave, 100 * v / tot))
print('%-32s %8.0f cycles/frame' % ('total (stamped)', tot / frames_per_wave))
