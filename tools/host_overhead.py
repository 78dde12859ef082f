#!/usr/bin/env python
"""Host-side cost of one fused pipeline call: steps/s on a tiny input (the kernel is then a few microseconds, so the
loop is bound by Python + ctypes + the lazy-fusion bookkeeping), next to the same loop on the bench workload."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).cuda()
for shape in ((1, 1, 4096), (256, 1, 160000)):
    x = torch.rand(*shape, device='cuda') * 2 - 1
    for _ in range(20):
        y = model(x)
    torch.cuda.synchronize()
    n = 2000 if shape[0] == 1 else 300
    t0 = time.perf_counter()
    for _ in range(n):
        y = model(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-18s issue %.1f us/step   issue+drain %.1f us/step' % (shape, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
