#!/usr/bin/env python
"""rows x 1 x 160 000 samples through tac.planned, 300 calls: for rocprofv3 --kernel-trace --stats (kernel time of a tiny launch).
    python tools/r06/small_trace.py [rows]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512), tac.AmplitudeToDb()).cuda()
x = torch.rand(rows, 1, 160000, device='cuda') * 2 - 1
fast = tac.planned(model, x)
for _ in range(300): fast(x)
torch.cuda.synchronize()
