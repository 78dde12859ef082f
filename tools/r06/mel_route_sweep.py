#!/usr/bin/env python
"""Which kernels the reference idiom Sequential(*Melspectrogram(...), AmplitudeToDb()) launches, and how long a call takes, over fft_lengths x
mel banks (256 rows x 160 000 samples, hop = fft_length / 4; 400: hop 160).   python tools/r06/mel_route_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
for n_fft in [int(v) for v in sys.argv[1:]] or (256, 400, 512, 1024, 2048, 4096):
    hop = 160 if n_fft == 400 else n_fft // 4
    for n_mels, sr in ((40, 16000), (64, 16000), (80, 16000), (80, 22050), (128, 22050), (128, 44100)):
        m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=n_mels, sample_rate=sr, fft_length=n_fft, hop_length=hop), tac.AmplitudeToDb()).cuda()
        try:
            for _ in range(10): m(x)
        except Exception as e:
            print('%4d / %3d bands %5d Hz: %s' % (n_fft, n_mels, sr, str(e)[:100])); continue
        before = dict(tac._hip.launches)
        m(x)
        calls = {k.replace('tac_', ''): v - before.get(k, 0) for k, v in tac._hip.launches.items() if v != before.get(k, 0)}
        torch.cuda.synchronize()
        ts = []
        for _ in range(100):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); m(x); b.record(); ts.append((a, b))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) for a, b in ts)
        frames = 256 * (1 + 160000 // hop)
        print('%4d / %3d bands %5d Hz: median %.4f ms = %5.0f M frames/s | %s' % (n_fft, n_mels, sr, v[len(v) // 2], frames / v[len(v) // 2] / 1e3, calls))
