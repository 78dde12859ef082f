#!/usr/bin/env python
"""Gradient of Spectrogram(n, n / 4, power=2) w.r.t. the waveform at fft_lengths that are not powers of two (64 x 160 000
samples): forward + backward wall time per step with the generic Stockham adjoint (csrc/stft_smooth.hip) and with the
DFT-matrix adjoint it replaced (``smooth_fft_size`` patched off for the gradient's frame stage only)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import importlib, torch
tac = importlib.import_module('torchaudio-contrib_amd')
H = tac._hip


def step_ms(mod, x, n=10):
    for _ in range(3):
        x.grad = None
        mod(x).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        x.grad = None
        mod(x).sum().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


x = torch.randn(64, 160000, device='cuda', requires_grad=True)
keep = H.smooth_fft_size
for n in (480, 960, 1200, 1920, 3000):
    mod = tac.Spectrogram(n, n // 4, power=2.).cuda()
    new = step_ms(mod, x)
    fg = H._frame_gradients

    def old_frames(gs, window, g, fg=fg):
        H.smooth_fft_size = lambda n_fft: False
        try:
            return fg(gs, window, g)
        finally:
            H.smooth_fft_size = keep
    H._frame_gradients = old_frames
    old = step_ms(mod, x)
    H._frame_gradients = fg
    print('n_fft %5d  fwd+bwd %7.3f ms with the Stockham adjoint, %7.3f ms with the DFT-matrix adjoint' % (n, new, old), flush=True)
