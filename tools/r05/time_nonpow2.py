#!/usr/bin/env python
"""fft_lengths that are not powers of two (256 x 160 000 samples, hop = N / 4): the float32 route against the float64
kernels (csrc/chain_f64.hip: generic Stockham passes of radix 4 / 2 / 3 / 5) and torch.stft on the same box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import importlib, torch
tac = importlib.import_module('torchaudio-contrib_amd')


def ms_of(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


x = torch.randn(256, 160000, device='cuda')
xd = x.double()
for n in (400, 480, 600, 800, 960, 1000, 1200, 1536, 1920, 2400, 3000, 6000, 2048):
    hop = n // 4
    frames = 1 + 160000 // hop
    gb = 256 * frames * (4 * hop + 8 * (n // 2 + 1)) / 1e9
    win = torch.hann_window(n, device='cuda')
    m32 = ms_of(lambda: tac.stft(x, n, hop))
    route = tac._native.lib().tac_last_route().decode()
    m64 = ms_of(lambda: tac.stft(xd, n, hop))
    mt = ms_of(lambda: torch.stft(x, n, hop, window=win, return_complex=True), 5)
    print('n_fft %5d  f32 %8.3f ms (%5.1f %% of 8 TB/s)   f64 kernels %8.3f ms   torch.stft f32 %8.3f ms' % (n, m32, gb / m32 * 1e3 / 80, m64, mt), flush=True)
