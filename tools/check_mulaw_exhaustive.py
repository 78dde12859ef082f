#!/usr/bin/env python
"""Exhaustive device-side check of the n_quantize = 256 mu-law encoder: every float32 in [-1, 1] (2 x 1 065 353 217
bit patterns) against a binary search in the reference's threshold tables (tests/golden g5 / _mulaw_tables.py).
    python tools/check_mulaw_exhaustive.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd import _mulaw_tables as tab

dev = torch.device('cuda')
pos = torch.tensor(tab.THR256_POS, dtype=torch.int64, device=dev)
neg = torch.tensor(tab.THR256_NEG, dtype=torch.int64, device=dev)
top, step = 0x3f800000, 1 << 27
bad = 0
for start in range(0, top + 1, step):
    bits = torch.arange(start, min(top + 1, start + step), dtype=torch.int64, device=dev)
    for sign, thr in ((0, pos), (1, neg)):
        raw = (bits | (sign << 31)).to(torch.int32) if sign == 0 else (bits - (1 << 31)).to(torch.int32)
        x = raw.view(torch.float32)
        got = tac.mu_law_encoding(x, 256)
        cnt = torch.searchsorted(thr, bits, right=True)
        want = tab.ZERO_CODE_256 + cnt if sign == 0 else tab.ZERO_CODE_256 - cnt
        bad += int((got != want).sum())
print('mismatches over 2 x %d float32 values: %d' % (top + 1, bad))
sys.exit(0 if bad == 0 else 1)
