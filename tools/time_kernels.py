#!/usr/bin/env python
"""Time the cfg-2 kernels with HIP events: python tools/time_kernels.py [stft spec mel ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac  # noqa: E402

which = sys.argv[1:] or ['stft', 'spec', 'mel']
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
fns = {}
stft_layer = tac.STFT(2048, 512).cuda()
fns['stft'] = lambda: tac.realize(stft_layer(x))
spec = tac.Spectrogram(2048, 512, power=2.).cuda()
fns['spec'] = lambda: spec(x)
specdb = torch.nn.Sequential(*tac.Spectrogram(2048, 512, power=1.), tac.AmplitudeToDb()).cuda()
fns['specdb'] = lambda: tac.realize(specdb(x))
mel = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                          tac.AmplitudeToDb()).cuda()
fns['mel'] = lambda: tac.realize(mel(x))
x4 = torch.rand(8, 8, 480000, device='cuda') * 2 - 1
s4 = tac.Spectrogram(4096, 1024).cuda()
fns['spec4096'] = lambda: s4(x4)
x5 = torch.rand(64, 1, 160000, device='cuda') * 2 - 1
s5 = tac.Spectrogram(512, 128, power=2.).cuda()
fns['spec512'] = lambda: s5(x5)
for name in which:
    fn = fns[name]
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n = 30
    st = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    en = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        st[i].record(); fn(); en[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(st, en))
    print('%s %-9s median %.4f ms  min %.4f ms' % (os.environ.get('TAC_AMD_LIB', 'default')[-24:], name, ts[n // 2], ts[0]), flush=True)
