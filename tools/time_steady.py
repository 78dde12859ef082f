#!/usr/bin/env python
"""Steady-state kernel times (0.5 s spin-up per kernel, then 100 back-to-back launches with per-launch HIP events).
    python tools/time_steady.py [stft spec mel]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
# TAC_ROTATE=n: the cfg-2 kernels (stft / spec / mel) visit n distinct input batches round-robin (n x 164 MB: beyond the 256 MiB
# Infinity Cache from n = 2), like bench.py
_xs = [x] + [torch.rand(256, 1, 160000, device='cuda') * 2 - 1 for _ in range(int(os.environ.get('TAC_ROTATE', '1')) - 1)]
_rot = [0]
def xr():
    _rot[0] = (_rot[0] + 1) % len(_xs)
    return _xs[_rot[0]]
stft = tac.STFT(2048, 512).cuda()
spec = tac.Spectrogram(2048, 512, power=2.).cuda()
mel = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                          tac.AmplitudeToDb()).cuda()
x4 = torch.rand(8, 8, 480000, device='cuda') * 2 - 1                       # cfg-4 slice: 4096/1024, magnitude
stft4 = tac.STFT(4096, 1024).cuda()
spec4 = tac.Spectrogram(4096, 1024).cuda()
x5 = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
stft512, spec512 = tac.STFT(512, 128).cuda(), tac.Spectrogram(512, 128, power=2.).cuda()
stft1k, spec1k = tac.STFT(1024, 256).cuda(), tac.Spectrogram(1024, 256, power=2.).cuda()
mel512 = torch.nn.Sequential(*tac.Melspectrogram(num_mels=80, sample_rate=16000, fft_length=512, hop_length=128),
                             tac.AmplitudeToDb()).cuda()
mel1k = torch.nn.Sequential(*tac.Melspectrogram(num_mels=80, sample_rate=16000, fft_length=1024, hop_length=256),
                            tac.AmplitudeToDb()).cuda()
mel400 = torch.nn.Sequential(*tac.Melspectrogram(num_mels=80, sample_rate=16000, fft_length=400, hop_length=160),
                             tac.AmplitudeToDb()).cuda()
stft400, spec400 = tac.STFT(400, 160).cuda(), tac.Spectrogram(400, 160, power=2.).cuda()
mel256 = torch.nn.Sequential(*tac.Melspectrogram(num_mels=40, sample_rate=8000, fft_length=256, hop_length=64),
                             tac.AmplitudeToDb()).cuda()                   # the three-phase band-sparse kernel
mel4k = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=48000, fft_length=4096, hop_length=1024),
                            tac.AmplitudeToDb()).cuda()
stft256, spec256 = tac.STFT(256, 64).cuda(), tac.Spectrogram(256, 64, power=2.).cuda()
fns = {'stft256': lambda: tac.realize(stft256(x5)), 'spec256': lambda: spec256(x5), 'mel4096': lambda: tac.realize(mel4k(x4)), 'mel256': lambda: tac.realize(mel256(x5)), 'mel400': lambda: tac.realize(mel400(x5)), 'stft400': lambda: tac.realize(stft400(x5)), 'spec400': lambda: spec400(x5),
       'stft': lambda: tac.realize(stft(xr())), 'spec': lambda: spec(xr()), 'mel': lambda: tac.realize(mel(xr())),
       'stft4096': lambda: tac.realize(stft4(x4)), 'spec4096': lambda: spec4(x4),
       'stft512': lambda: tac.realize(stft512(x5)), 'spec512': lambda: spec512(x5),
       'stft1024': lambda: tac.realize(stft1k(x5)), 'spec1024': lambda: spec1k(x5),
       'mel512': lambda: tac.realize(mel512(x5)), 'mel1024': lambda: tac.realize(mel1k(x5))}
for name in (sys.argv[1:] or ['stft', 'spec', 'mel']):
    fn = fns[name]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    n = 100
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print('%-10s %-5s median %.4f ms  p10 %.4f  p90 %.4f' % (os.environ.get('TAC_AMD_LIB', 'default')[-14:], name, ts[n // 2], ts[n // 10], ts[9 * n // 10]))
