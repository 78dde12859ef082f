#!/usr/bin/env python
"""cfg-2 forward chain: K eager launches vs the same K launches captured once in a HIP graph and replayed (four rotating input batches).
    python tools/r04/graph_replay.py [K]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda', 0)
model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512), tac.AmplitudeToDb()).to(dev)
xs = [torch.rand(256, 1, 160000, device=dev) * 2 - 1 for _ in range(4)]
frames = 256 * (1 + 160000 // 512)


def eager():
    y = None
    for i in range(K):
        y = tac.realize(model(xs[i % 4]))
    return y


for _ in range(3):
    eager()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eager()
side.synchronize()
with torch.cuda.graph(g):
    yg = eager()
torch.cuda.synchronize()


def timed(fn, reps=25):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K)
    out.sort()
    return out[len(out) // 2], out[2], out[-3]


for r in range(2):
    e = timed(eager)
    h = timed(g.replay)
    print('eager  %.4f ms/step (p10 %.4f p90 %.4f) = %.1f M frames/s' % (e[0] * 1e3, e[1] * 1e3, e[2] * 1e3, frames / e[0] / 1e6))
    print('graph  %.4f ms/step (p10 %.4f p90 %.4f) = %.1f M frames/s' % (h[0] * 1e3, h[1] * 1e3, h[2] * 1e3, frames / h[0] / 1e6))
want = tac.realize(model(xs[(K - 1) % 4]))
print('last replayed result equals the eager one:', torch.equal(yg, want))
