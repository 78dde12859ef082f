#!/usr/bin/env python
"""STFT (complex rows) and Spectrogram (|X|^2 rows) over fft_lengths x hops on 256 rows x 160 000 samples: time, algorithmic GB/s (4 hop + 8 F /
4 hop + 4 F bytes per frame) and the kernel the library names (tac_last_route where it is recorded).   python tools/r06/stft_route_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
def timed(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts)
    return v[len(v) // 2]
for n_fft in (256, 400, 512, 1024, 2048, 4096):
    for hop in sorted({n_fft // 8, n_fft // 4, n_fft // 2, n_fft // 4 + 4, 160}):
        if hop > n_fft: continue
        T, F = 1 + 160000 // hop, n_fft // 2 + 1
        for name, fn, per in (('stft', lambda: tac.realize(tac.stft(x, n_fft, hop)), 4 * hop + 8 * F),
                              ('spec', lambda: tac.realize(tac.Spectrogram(n_fft, hop, power=2.).cuda()(x)), 4 * hop + 4 * F)):
            try:
                sp = tac.Spectrogram(n_fft, hop, power=2.).cuda()
                f = (lambda: tac.realize(tac.stft(x, n_fft, hop))) if name == 'stft' else (lambda: tac.realize(sp(x)))
                ms = timed(f)
            except Exception as e:
                print('%4d / %4d %s: %s' % (n_fft, hop, name, str(e)[:80])); continue
            gbs = 256 * T * per / ms / 1e6
            print('%4d / %4d %s: %.4f ms  %5.0f GB/s = %4.1f %% of 8 TB/s' % (n_fft, hop, name, ms, gbs, gbs / 80))
