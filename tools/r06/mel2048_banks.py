#!/usr/bin/env python
"""The reference idiom at fft_length 2048 / hop 512 on cfg-2's 256 x 160 000 samples for the usual mel banks: launches per call and time
(TAC_AMD_LIB selects the library build).   python tools/r06/mel2048_banks.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
for n_mels, sr in ((40, 16000), (64, 16000), (80, 16000), (80, 22050), (96, 16000), (100, 22050), (128, 16000), (128, 44100), (160, 16000), (256, 48000)):
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=n_mels, sample_rate=sr, fft_length=2048, hop_length=512), tac.AmplitudeToDb()).cuda()
    for _ in range(20): m(x)
    before = dict(tac._hip.launches)
    m(x)
    calls = {k: v - before.get(k, 0) for k, v in tac._hip.launches.items() if v != before.get(k, 0)}
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m(x); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts)
    pack = tac._hip._melbank_pack(m[2].filterbank, 2048)
    print('%3d bands %5d Hz: median %.4f ms | %s | pack info %s' % (n_mels, sr, v[len(v) // 2], calls, None if pack is None else [int(t) for t in pack[2]]))
