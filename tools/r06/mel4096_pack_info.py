#!/usr/bin/env python
"""What tac_melbank_pack builds for fft_length 4096 (the piece layout of csrc/stft_n4096_s3.hpp) on the usual mel banks, and the time of
the one-launch chain with each (8 x 8 x 480 000 samples).   python tools/r06/mel4096_pack_info.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
x = torch.rand(8, 8, 480000, device='cuda') * 2 - 1
for n_mels, sr, htk in ((40, 16000, False), (64, 22050, False), (80, 44100, False), (80, 48000, True), (96, 44100, False), (128, 44100, False),
                        (128, 48000, False), (128, 16000, False), (200, 48000, False), (256, 48000, False)):
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=n_mels, sample_rate=sr, fft_length=4096, hop_length=1024, htk=htk), tac.AmplitudeToDb()).cuda()
    fb = m[2].filterbank
    pack = tac._hip._melbank_pack(fb, 4096)
    info = None if pack is None else [int(v) for v in pack[2]]
    for _ in range(20): m(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m(x); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts)
    print('%3d bands %5d Hz htk=%d: info [floats, slots, mark, steps, waves, pieces, rounds] = %s | median %.4f ms' % (n_mels, sr, htk, info, v[len(v) // 2]))
