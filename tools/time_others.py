#!/usr/bin/env python
"""Steady-state timings of the standalone kernels (not on the fused path) at cfg-2 / cfg-5 sizes, with the bytes each
must move and the resulting fraction of the 8 TB/s HBM peak."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac


def steady(fn, n=60):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[n // 2]


x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
z = tac.realize(tac.STFT(2048, 512).cuda()(x))                      # (256,1,1025,313,2) strided view
p = tac.Spectrogram(2048, 512, power=2.).cuda()(x)                  # (256,1,1025,313)
pc = p.contiguous()
fb = tac.create_mel_filter(1025, 128, 0.0, 8000.0, False).cuda()
adv = torch.linspace(0, math.pi * 512, 1025)[..., None].cuda()
xm = torch.rand(1024, 1, 120000, device='cuda') * 2 - 1
codes = tac.mu_law_encoding(xm, 256)
stft400 = tac.STFT(400, 160).cuda()
spec400 = tac.Spectrogram(400, 160, power=2.).cuda()
x4k = torch.rand(64, 1, 480000, device='cuda') * 2 - 1
p4k = tac.Spectrogram(4096, 1024, power=2.).cuda()(x4k)             # (64,1,2049,469)
fb4k = tac.create_mel_filter(2049, 128, 0.0, 22050.0, False).cuda()
cases = [
    ('STFT n_fft=400 hop=160 (mixed radix)', lambda: tac.realize(stft400(x)), x.numel() * 4 + 256 * 1001 * 201 * 8),
    ('Spectrogram n_fft=400 (mixed radix)', lambda: spec400(x), x.numel() * 4 + 256 * 1001 * 201 * 4),
    ('complex_norm (power 2)', lambda: tac.complex_norm(z, 2.0), z.numel() * 4 + z.numel() * 2),
    ('magphase', lambda: tac.magphase(z, 1.0), z.numel() * 4 + z.numel() * 4),
    ('apply_filterbank 1025x128', lambda: tac.apply_filterbank(p, fb), p.numel() * 4 + p.numel() // 1025 * 128 * 4),
    ('apply_filterbank 2049x128 (4096 rows)', lambda: tac.apply_filterbank(p4k, fb4k), p4k.numel() * 4 + p4k.numel() // 2049 * 128 * 4),
    ('amplitude_to_db', lambda: tac.amplitude_to_db(p), p.numel() * 8),
    ('db_to_amplitude', lambda: tac.db_to_amplitude(p), p.numel() * 8),
    ('phase_vocoder rate 1.3', lambda: tac.phase_vocoder(z, 1.3, adv), z.numel() * 4 + int(z.numel() / 1.3) * 4),
    ('mu_law_encoding (cfg-5)', lambda: tac.mu_law_encoding(xm, 256), xm.numel() * 12),
    ('mu_law_decoding (cfg-5)', lambda: tac.mu_law_decoding(codes, 256), xm.numel() * 12),
    ('hpss k=31 (frame-major |X|^2)', lambda: tac.hpss(p, 31, 2.0), p.numel() * 20),
    ('hpss k=31, mask_only', lambda: tac.hpss(p, 31, 2.0, False, True), p.numel() * 12),
    ('hpss k=17 (frame-major |X|^2)', lambda: tac.hpss(p, 17, 2.0), p.numel() * 20),
    ('hpss k=31 (contiguous)', lambda: tac.hpss(pc, 31, 2.0), p.numel() * 20),
    ('hpss k=(5, 9) (two launches)', lambda: tac.hpss(pc, (5, 9), 2.0), p.numel() * 20),
]
only = sys.argv[1:]
for name, fn, nbytes in cases:
    if only and not any(o in name for o in only):
        continue
    ms = steady(fn)
    print('%-38s %.4f ms   %7.1f MB   %5.2f TB/s  (%.0f %% of 8 TB/s)' % (name, ms, nbytes / 1e6, nbytes / ms / 1e9, 100 * nbytes / ms / 1e9 / 8))
