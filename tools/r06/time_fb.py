#!/usr/bin/env python
"""The dense fp32 MFMA filterbank GEMM (gemm_fb_kernel) at cfg-2: 80 128 frames x 1025 bins x 128 bands, frame-major spectrogram.
    python tools/r06/time_fb.py [lib.so ...]      (default: the package's library; several: same-process A/B)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac

P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
paths = sys.argv[1:] or [tac._native.LIB_PATH]
libs = []
for pth in paths:
    h = ctypes.CDLL(os.path.abspath(pth))
    h.tac_apply_filterbank_f32.argtypes = [P, I64, I32, I64, I64, I64, I64, P, P, I32, P, P]
    libs.append((os.path.basename(pth), h))
dev = torch.device('cuda', 0)
rows, T, F, M = 256, 313, 1025, 128
spec = torch.rand(rows, T, F, device=dev)
fb = torch.rand(F, M, device=dev)
out = torch.empty(rows, T, M, device=dev)
stream = P(torch.cuda.current_stream().cuda_stream)
want = (spec.reshape(-1, F).double() @ fb.double()).float().reshape(rows, T, M)
flops = 2.0 * rows * T * F * M
for name, h in libs:
    def launch():
        rc = h.tac_apply_filterbank_f32(P(spec.data_ptr()), rows, F, T, T * F, 1, F, P(fb.data_ptr()), None, M, P(out.data_ptr()), stream)
        assert rc == 0, rc
    out.zero_(); launch(); torch.cuda.synchronize()
    err = ((out - want).abs().max() / want.abs().max()).item()
    for _ in range(20): launch()
    ts = []
    for rnd in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): launch()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 50)
    ts.sort()
    print('%-28s max rel err %.2g | median %.4f ms (min %.4f) = %.1f TFLOP/s = %.1f %% of the 157.3 TFLOP/s f32 MFMA peak'
          % (name, err, ts[3], ts[0], flops / ts[3] / 1e9, 100 * flops / ts[3] / 1e9 / 157.3))
