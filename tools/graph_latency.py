#!/usr/bin/env python
"""Launch-bound regime (streaming inference: one short utterance per call): eager fused call vs the same call
captured once in a HIP graph and replayed.  The steady-state path allocates only through torch's caching allocator
and launches one kernel on the current stream, so it is capturable after a warm-up call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac

model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).cuda()
for shape in ((1, 1, 16000), (8, 1, 16000), (256, 1, 160000)):
    x = torch.rand(*shape, device='cuda') * 2 - 1
    for _ in range(5):
        ref = tac.realize(model(x))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = tac.realize(model(x))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref), 'graph replay differs from the eager result'
    x.copy_(torch.rand(*shape, device='cuda') * 2 - 1)      # new input in the captured buffer
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, tac.realize(model(x)))
    n = 2000 if shape[0] < 256 else 200
    t0 = time.perf_counter()
    for _ in range(n):
        tac.realize(model(x))
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n * 1e6
    print('%-18s eager %.1f us/call   graph replay %.1f us/call' % (shape, eager, graph))
