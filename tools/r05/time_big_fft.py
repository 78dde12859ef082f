#!/usr/bin/env python
"""Timing of the four-step kernel (csrc/stft_big.hip) at fft_length 8192 / 16384 / 32768, hop = fft_length / 4, 16 rows of
2 880 000 samples: complex rows and |X|^2 rows, algorithmic bytes (4 hop + 8 or 4 x bins per frame) over the event time,
torch.stft (hipFFT + its framing copies) beside it.  Also the HPSS widths added in round 5."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import importlib, torch
tac = importlib.import_module('torchaudio-contrib_amd')


def ms_of(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


x = torch.randn(16, 1, 2880000, device='cuda')
for n in (4096, 8192, 16384, 32768):
    hop = n // 4
    frames = 1 + 2880000 // hop
    win = torch.hann_window(n, device='cuda')
    for name, fn, per in (('complex', lambda: tac.stft(x, n, hop), 4 * hop + 8 * (n // 2 + 1)),
                          ('power', tac.Spectrogram(n, hop, power=2.).cuda(), 4 * hop + 4 * (n // 2 + 1))):
        f = fn if name == 'complex' else (lambda m=fn: m(x))
        ms = ms_of(f)
        gb = 16 * frames * per / 1e9
        print('n_fft %5d %-8s %7.3f ms  %6.1f GB/s alg (%4.1f %% of 8 TB/s)' % (n, name, ms, gb / ms * 1e3, gb / ms * 1e3 / 80), flush=True)
    ms = ms_of(lambda: torch.stft(x.reshape(16, -1), n, hop, window=win, return_complex=True), 5)
    print('n_fft %5d torch.stft %7.3f ms' % (n, ms), flush=True)
s = torch.rand(8, 1025, 5000, device='cuda')
for k in (31, (31, 17), 33, 47, 63, (63, 33)):
    ms = ms_of(lambda: tac.hpss(s, k, 2.0, False), 10)
    print('hpss 8 x 1025 x 5000 width %-9s %7.3f ms' % (k, ms), flush=True)
