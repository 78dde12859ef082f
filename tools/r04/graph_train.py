#!/usr/bin/env python
"""cfg-2 training step (forward + backward of Sequential(*Melspectrogram, AmplitudeToDb)): eager autograd vs
torch.cuda.make_graphed_callables (forward and backward each captured once in a HIP graph and replayed)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac

dev = torch.device('cuda', 0)
model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512), tac.AmplitudeToDb()).to(dev)
x = (torch.rand(256, 1, 160000, device=dev) * 2 - 1).requires_grad_(True)
K = 60


def step(m):
    y = tac.realize(m(x))
    y.backward(torch.ones_like(y))
    g = x.grad
    x.grad = None
    return g


def timed(m, reps=9):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step(m)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K)
    out.sort()
    return out[len(out) // 2]


g_eager = step(model).clone()
for _ in range(5):
    step(model)
print('eager   %.4f ms per training step' % (timed(model) * 1e3))
try:
    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, t):
            return tac.realize(self.m(t))
    graphed = torch.cuda.make_graphed_callables(Wrap(model), (x,))
    g_graph = step(graphed).clone()
    print('graphed %.4f ms per training step; gradient equal to eager: %s (max |diff| %.3g)'
          % (timed(graphed) * 1e3, torch.equal(g_graph, g_eager), (g_graph - g_eager).abs().max().item()))
except Exception as e:                                     # noqa: BLE001
    print('make_graphed_callables failed: %s: %s' % (type(e).__name__, str(e)[:300]))
