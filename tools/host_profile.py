#!/usr/bin/env python
"""cProfile of the host side of one fused pipeline call on a tiny input (where the kernel is negligible)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).cuda()
x = torch.rand(1, 1, 4096, device='cuda') * 2 - 1
for _ in range(50):
    model(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    model(x)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
