#!/usr/bin/env python
"""Same-process A/B of library builds through the C ABI (ctypes, no torch dispatch): every build is loaded side by side and the
launches alternate A B C A B C ... on the SAME device buffers, so box, clock drift and buffer placement are shared.
    python tools/r04/ab_inproc.py <op> name=path [name=path ...]      op: stft | spec | mel | bwd (fused mel-chain backward)
cfg-2 (256 x 160 000 samples, 2048 / 512, 128 mel + dB), TAC_ROTATE distinct input batches (default 4), N launches per build
(TAC_AB_N, default 300).  Prints median / p10 / p90 per build, and the median of per-round differences to the first build."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd._native import StftDesc

op = sys.argv[1]
builds = [a.split('=', 1) for a in sys.argv[2:]]
P, I32, F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
libs = []
for name, path in builds:
    h = ctypes.CDLL(os.path.abspath(path))
    h.tac_stft_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), P, P]
    h.tac_spectrogram_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), F, ctypes.c_int, F, F, P, P]
    h.tac_melbank_pack.argtypes = [P, I32, I32, I32, P, I32, P, I32, P, P]
    h.tac_melspec_sparse_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), F, P, P, P, I32, ctypes.c_int, F, F, P, P]
    h.tac_filterbank_adjoint_pack.argtypes = [P, I32, I32, P, P, P]
    h.tac_spectrogram_backward_ola_workspace.argtypes = [ctypes.POINTER(StftDesc)]
    h.tac_spectrogram_backward_ola_workspace.restype = ctypes.c_int64
    h.tac_melspectrogram_backward_ola_f32.argtypes = [P, P, ctypes.POINTER(StftDesc), P, I32, P, I32, F, P, ctypes.c_int64, P, ctypes.c_int64, P]
    libs.append((name, h))
dev = torch.device('cuda', 0)
nrot = int(os.environ.get('TAC_ROTATE', '4'))
N = int(os.environ.get('TAC_AB_N', '300'))
rows, L, n_fft, hop, n_mels = 256, 160000, 2048, int(os.environ.get('TAC_AB_HOP', '512')), 128
T = 1 + L // hop
xs = [torch.rand(rows, L, device=dev) * 2 - 1 for _ in range(nrot)]
window = torch.hann_window(n_fft, device=dev)
fb = tac.create_mel_filter(n_fft // 2 + 1, n_mels, 0.0, 8000.0, False).to(dev).contiguous()
desc = StftDesc(rows, L, L, n_fft, hop, n_fft, 1, 1, 0, 1, 0)
stream = P(torch.cuda.current_stream().cuda_stream)
width = {'stft': 2 * 1025, 'spec': 1025, 'mel': n_mels, 'bwd': 0}[op]
out = torch.empty(rows * T * width if op != 'bwd' else rows * L, device=dev)
packs = {}
if op == 'mel':
    for name, h in libs:
        wpack = torch.empty(24576, device=dev); dsc = torch.empty(4096, dtype=torch.int32, device=dev); info = (I32 * 8)()
        # a build named "...+p" is timed on the PIECE layout of the bank (tac_melbank_pack with n_fft = -2048), others on the classic one
        rc = h.tac_melbank_pack(P(fb.data_ptr()), 1025, n_mels, -2048 if name.endswith('+p') else n_fft, P(wpack.data_ptr()), 24576, P(dsc.data_ptr()), 4096, ctypes.cast(info, P), stream)
        assert rc == 0, (name, rc)
        packs[name] = (wpack, dsc, info)


bwd = {}
if op == 'bwd':
    gmel = torch.rand(rows * T * n_mels, device=dev)
    for name, h in libs:
        table = torch.empty(16 * 1025 + 16, dtype=torch.uint8, device=dev); mx = (I32 * 1)()
        assert h.tac_filterbank_adjoint_pack(P(fb.data_ptr()), 1025, n_mels, P(table.data_ptr()), ctypes.cast(mx, P), stream) == 0 and mx[0] <= 2
        nbytes = h.tac_spectrogram_backward_ola_workspace(ctypes.byref(desc))
        assert nbytes > 0, (name, nbytes)
        bwd[name] = (table, torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes)


def launch(name, h, x):
    if op == 'stft':
        rc = h.tac_stft_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), P(out.data_ptr()), stream)
    elif op == 'spec':
        rc = h.tac_spectrogram_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), 2.0, 0, 1.0, 1e-7, P(out.data_ptr()), stream)
    elif op == 'bwd':
        table, ws, nbytes = bwd[name]
        rc = h.tac_melspectrogram_backward_ola_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), P(gmel.data_ptr()), n_mels,
                                                   P(table.data_ptr()), 1025, 2.0, P(ws.data_ptr()), nbytes, P(out.data_ptr()), L, stream)
    else:
        wpack, dsc, info = packs[name]
        rc = h.tac_melspec_sparse_f32(P(x.data_ptr()), P(window.data_ptr()), ctypes.byref(desc), 2.0, P(wpack.data_ptr()), P(dsc.data_ptr()),
                                      ctypes.cast(info, P), n_mels, 1, 1.0, 1e-7, P(out.data_ptr()), stream)
    assert rc == 0, (name, rc)


# results agree between builds (a quick guard against timing a broken variant)
ref = None
for name, h in libs:
    launch(name, h, xs[0]); torch.cuda.synchronize()
    cur = out.clone()
    if ref is None: ref = cur
    else:
        d = (cur - ref).abs().max().item()
        print('check %-10s max |diff| vs %s: %.3g' % (name, libs[0][0], d))
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 1.0:
    for name, h in libs:
        launch(name, h, xs[k % nrot]); k += 1
    torch.cuda.synchronize()
ev = {name: [] for name, _ in libs}
for r in range(N):
    order = libs if r % 2 == 0 else libs[::-1]                     # alternate the order inside a round as well
    for name, h in order:                                           # every launch takes the next batch (never the one just read)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(name, h, xs[k % nrot]); b.record(); k += 1
        ev[name].append((a, b))
    if nrot % len(libs) == 0:
        k += 1                                                      # ... and every build meets every batch
torch.cuda.synchronize()
ts = {name: [a.elapsed_time(b) for a, b in v] for name, v in ev.items()}
q = lambda v, f: sorted(v)[int(f * (len(v) - 1))]
base = libs[0][0]
for name, _ in libs:
    v = ts[name]
    diff = sorted(x - y for x, y in zip(v, ts[base]))
    print('%-4s %-12s median %.4f ms  p10 %.4f  p90 %.4f  mean %.4f | vs %s: median of per-round differences %+.4f ms (%+.2f %%)'
          % (op, name, q(v, .5), q(v, .1), q(v, .9), sum(v) / len(v), base, q(diff, .5), 100 * q(diff, .5) / q(ts[base], .5)))
