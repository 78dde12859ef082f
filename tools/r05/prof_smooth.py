#!/usr/bin/env python
"""Three launches of the generic Stockham kernel (fft_length 960 / hop 240, 256 x 160 000 samples) for counter passes:
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -- python tools/r05/prof_smooth.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import importlib, torch
tac = importlib.import_module('torchaudio-contrib_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 960
x = torch.randn(256, 160000, device='cuda')
for _ in range(3):
    y = tac.stft(x, n, n // 4)
torch.cuda.synchronize()
