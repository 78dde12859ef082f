#!/usr/bin/env python
"""Tiny driver for rocprofv3: launch one kernel family a few times at cfg-2 size.
    python tools/prof_driver.py mel|stft|spec|fb|grad|gradf|gradspec|mulaw [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'mel'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
if what == 'mel':
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).cuda()
    fn = lambda: m(x)
elif what == 'stft':
    layer = tac.STFT(2048, 512).cuda()
    fn = lambda: tac.stft(x, 2048, 512)
elif what == 'spec':
    s = tac.Spectrogram(2048, 512, power=2.).cuda()
    fn = lambda: s(x)
elif what == 'unfused':
    tac.set_lazy_fusion(False)
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).cuda()
    fn = lambda: m(x)
elif what == 'fb':
    # the filterbank stage as a dense fp32 MFMA GEMM (a random bank is not band-sparse): north_star's MFMA-utilisation figure
    spec = tac.Spectrogram(2048, 512, power=2.).cuda()(x)
    fbd = torch.rand(1025, 128, device='cuda')
    fn = lambda: tac.apply_filterbank(spec, fbd)
elif what == 'grad':
    # forward + backward of the reference idiom with a waveform that requires grad: deferred like any other call (one
    # fused forward kernel), differentiated through the tac_amd::melspectrogram op's HIP gradient kernels
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).cuda()
    xg = x.clone().requires_grad_(True)
    def fn():
        y = m(xg)
        y.backward(torch.ones_like(y))
        return y
elif what == 'spec4096':
    # cfg-4 slice: 64 rows x 480 000 samples, fft_length 4096 / hop 1024, magnitude rows
    x4 = torch.rand(8, 8, 480000, device='cuda') * 2 - 1
    m = tac.Spectrogram(4096, 1024).cuda()
    fn = lambda: m(x4)
elif what == 'mel4096':
    # the one-launch chain at fft_length 4096 (44.1 kHz bank, 128 bands + dB) on the cfg-4 slice
    x4 = torch.rand(8, 8, 480000, device='cuda') * 2 - 1
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=44100, fft_length=4096, hop_length=1024),
                            tac.AmplitudeToDb()).cuda()
    fn = lambda: m(x4)
elif what == 'gradspec':
    # training step through the Spectrogram layer (power 2): fused forward kernel, one backward kernel + border fold
    m = tac.Spectrogram(2048, 512, power=2.).cuda()
    xg = x.clone().requires_grad_(True)
    def fn():
        y = m(xg)
        y.backward(torch.ones_like(y))
        return y
elif what == 'gradf':
    # the same through the factory container (one tac_amd::melspectrogram op): fused forward kernel, and a backward whose
    # inverse-FFT kernel forms the gradient spectrum on load
    m = torch.nn.Sequential(tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).cuda()
    xg = x.clone().requires_grad_(True)
    def fn():
        y = m(xg)
        y.backward(torch.ones_like(y))
        return y
elif what.startswith('grad') and what[4:].split('x')[0].split('h')[0].isdigit():
    # grad1024 / grad512: training step of the 80-band chain at that fft_length, hop = n / 4 (overlap-add in LDS);
    # grad1024x: hop = n / 4 - 6, which is not a multiple of n / 16: frame gradients through memory + gather overlap-add
    # grad400h160: explicit hop
    n = int(what[4:].split('x')[0].split('h')[0])
    hop = int(what.split('h')[1]) if 'h' in what[4:] else n // 4 - (6 if what.endswith('x') else 0)
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=80, sample_rate=16000, fft_length=n, hop_length=hop),
                            tac.AmplitudeToDb()).cuda()
    xg = x.clone().requires_grad_(True)
    def fn():
        y = m(xg)
        y.backward(torch.ones_like(y))
        return y
elif what == 'mulaw':
    xm = torch.rand(1024, 1, 120000, device='cuda') * 2 - 1
    fn = lambda: tac.mu_law_decoding(tac.mu_law_encoding(xm, 256), 256)
for _ in range(iters):
    y = fn()
torch.cuda.synchronize()
print(what, tuple(y.shape))
