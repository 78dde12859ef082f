#!/usr/bin/env python
"""Host time per call of the reference idiom on one row (bench.py stages.small_batch, quickly): the module chain and tac.planned."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512), tac.AmplitudeToDb()).cuda()
for rows in (1, 4):
    x = torch.rand(rows, 1, 160000, device='cuda') * 2 - 1
    fast = tac.planned(model, x)
    ended = torch.nn.Sequential(*list(model)[:-1])
    for name, fn in (('Sequential(*Melspectrogram, AmplitudeToDb)', lambda: model(x)), ('tac.planned(...)', lambda: fast(x)),
                     ('Sequential(*Melspectrogram) [chain end]', lambda: ended(x))):
        for _ in range(300): fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(2000): fn()
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            ta = time.perf_counter() - t0
            best = min(best, th)
        print('rows %d  %-46s host %.2f us per call (last round incl. drain %.2f), binding %s' % (rows, name, best / 2000 * 1e6, ta / 2000 * 1e6, tac._native.binding()))
