#!/usr/bin/env python
"""Debug: with a -DTAC_S3_CYCLES=1 build (TAC_AMD_LIB=...), shader cycles (s_memtime) and 100 MHz wall ticks of every wave's frame
loop of melspec_stream3_kernel at cfg-2, on one re-read batch and on TAC_ROTATE distinct batches: the same cycles at a lower
cycles-per-tick ratio = the HBM-resident penalty is shader clock, more cycles = it is latency."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                        tac.AmplitudeToDb()).cuda()
for nrot in (1, 4):
    xs = [torch.rand(256, 1, 160000, device='cuda') * 2 - 1 for _ in range(nrot)]
    import time
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 0.6:
        for _ in range(10):
            y = m(xs[k % nrot]); k += 1
        torch.cuda.synchronize()
    cyc, tick, setup = [], [], []
    for r in range(40):
        y = m(xs[k % nrot]); k += 1
        torch.cuda.synchronize()
        phys = y.transpose(-2, -1).contiguous().view(-1)[:256 * 12 * 2].view(256 * 12, 2).cpu()
        cyc.append(phys[:, 0].mean().item()); tick.append(phys[:, 1].mean().item())
        setup.append(y.transpose(-2, -1).contiguous().view(-1)[256 * 12 * 2:256 * 12 * 2 + 256].mean().item())
    fine = y.transpose(-2, -1).contiguous().view(-1)[256 * 12 * 2 + 256:256 * 12 * 2 + 260].cpu().tolist()
    print('  block 7 wave 0 (10 ns ticks): entry->loads issued %d, loads landed +%d, stores issued +%d, barrier + first request +%d' % tuple(fine))
    c = sum(cyc) / len(cyc); t = sum(tick) / len(tick)
    print('batches %d: %.0f shader cycles per wave, %.0f ticks of 10 ns (%.1f us) -> %.0f MHz; set-up before the loop %.2f us' % (nrot, c, t, t / 100.0, c / t * 100.0, sum(setup) / len(setup) / 100.0))
