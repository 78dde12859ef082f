#!/usr/bin/env python
"""Time series of a kernel under sustained back-to-back launches (does the power rows' slow mode come with sustained load?).
    python tools/r04/sustained.py spec|stft|mel [seconds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchaudio_contrib_amd as tac
what = sys.argv[1] if len(sys.argv) > 1 else 'spec'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
xs = [torch.rand(256, 1, 160000, device='cuda') * 2 - 1 for _ in range(4)]
if what == 'spec':
    m = tac.Spectrogram(2048, 512, power=2.).cuda(); fn = lambda x: m(x)
elif what == 'stft':
    fn = lambda x: tac.stft(x, 2048, 512)
else:
    m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512), tac.AmplitudeToDb()).cuda(); fn = lambda x: m(x)
for _ in range(20):
    fn(xs[0])
torch.cuda.synchronize()
t0 = time.perf_counter(); k = 0; series = []
while time.perf_counter() - t0 < secs:
    ev = []
    for _ in range(100):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(xs[k % 4]); b.record(); k += 1
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    series.append((time.perf_counter() - t0, ts[50], ts[10], ts[90]))
print(what, 'median per 100 launches over time:')
print(' '.join('%.1fs:%.4f' % (t, m) for t, m, _, _ in series))
