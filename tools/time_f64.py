"""float64 chain: the chain_f64.hip kernels beside torch's own GPU operators (rocFFT / rocBLAS) on cfg-2-shaped input.
Usage: python tools/time_f64.py  (on the GPU box)"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
tac = importlib.import_module('torchaudio-contrib_amd')
C = importlib.import_module('torchaudio-contrib_amd._composite')


def clock(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    x = torch.rand(256, 1, 160000, device='cuda', dtype=torch.float64) * 2 - 1
    for n_fft, hop, mels in ((2048, 512, 128), (512, 128, 80), (400, 160, 80)):
        win = torch.hann_window(n_fft, dtype=torch.float64, device='cuda')
        fb = tac.create_mel_filter(n_fft // 2 + 1, mels, 0.0, 8000.0, False).double().cuda()
        args = (n_fft, hop, n_fft, True, 'reflect', False, True)
        ours = clock(lambda: torch.ops.tac_amd.spectrogram(x, win, *args, 2.0, False, 1.0, 1e-7))
        theirs = clock(lambda: C.spectrogram(x, win, *args, 2.0, False, 1.0, 1e-7))
        print('spectrogram f64 %4d/%3d      kernels %.3f ms   torch ops %.3f ms' % (n_fft, hop, ours, theirs))
        ours = clock(lambda: torch.ops.tac_amd.melspectrogram(x, win, fb, *args, 2.0, True, 1.0, 1e-7))
        theirs = clock(lambda: C.melspectrogram(x, win, fb, *args, 2.0, True, 1.0, 1e-7))
        print('mel + dB    f64 %4d/%3d/%3d  kernels %.3f ms   torch ops %.3f ms' % (n_fft, hop, mels, ours, theirs))


if __name__ == '__main__':
    main()
