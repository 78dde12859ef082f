#!/usr/bin/env python
# NOTE (round 6): the timing / probe switches this tool builds with left the product sources; apply tools/ablation/lab_knobs_r06.patch
# (patch -p1 at the repo root) to a scratch tree first.
"""Diagnostic: shader cycles and wall time (100 MHz ticks) of the MEDIAN phase of hpss tiles, against the number of result arrays the
kernel stores (-DTAC_HPSS_PROBE -DTAC_HPSS_ABL_ARRAYS=n builds).  Same cycles + longer time = the clock dropped; more cycles = stalls.
    python tools/r04/hpss_probe.py name=lib.so ..."""
import ctypes, os, sys
import torch
P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
dev = torch.device('cuda', 0)
rows, F, T, k = 256, 1025, 313, int(os.environ.get('TAC_HPSS_K', '31'))
z = torch.rand(rows, T, F, device=dev) ** 2
outs = [torch.empty(rows, T, F, device=dev) for _ in range(4)]
stream = P(torch.cuda.current_stream().cuda_stream)
for a in sys.argv[1:]:
    name, path = a.split('=', 1)
    h = ctypes.CDLL(os.path.abspath(path))
    h.tac_hpss_f32.argtypes = [P, I64, I32, I32, I64, I64, I64, I32, I32, ctypes.c_float, ctypes.c_int, P, P, P, P, P]
    h.tac_debug_hpss_probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]

    def launch():
        rc = h.tac_hpss_f32(P(z.data_ptr()), rows, F, T, T * F, 1, F, k, k, 2.0, 0, P(outs[0].data_ptr()), P(outs[1].data_ptr()),
                            P(outs[2].data_ptr()), P(outs[3].data_ptr()), stream)
        assert rc == 0
    for _ in range(30):
        launch()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 4)()
    h.tac_debug_hpss_probe(buf, 1)
    n = 50
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a_, b_ in ev:
        a_.record(); launch(); b_.record()
    torch.cuda.synchronize()
    h.tac_debug_hpss_probe(buf, 0)
    cyc, ticks, cnt = buf[0], buf[1], buf[2]
    ms = sorted(x.elapsed_time(y) for x, y in ev)[n // 2]
    print('%-8s kernel %.4f ms | median phase per tile: %.0f shader cycles, %.2f us -> %.0f MHz' % (name, ms, cyc / cnt, ticks / cnt / 100.0, cyc / ticks * 100.0))
