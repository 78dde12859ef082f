#!/usr/bin/env python
"""Does the time of the STFT row kernels depend on WHERE their buffers lie?  cfg-2 power rows (and complex rows with 'stft')
through the C ABI on one input batch out of four rotating ones, the output placed at a series of byte offsets inside one big
slab (and the inputs likewise): median of 60 launches per placement.
    python tools/r04/placement_scan.py [spec|stft]"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd._native import StftDesc
op = sys.argv[1] if len(sys.argv) > 1 else 'spec'
h = tac._native.lib()
P = ctypes.c_void_p
dev = torch.device('cuda', 0)
rows, L, n_fft, hop = 256, 160000, 2048, 512
T = 1 + L // hop
width = 2050 if op == 'stft' else 1025
nrot = 4
out_elems = rows * T * width
slab_out = torch.empty(out_elems + (64 << 20) // 4, device=dev)
slab_in = torch.rand(nrot * rows * L + (64 << 20) // 4, device=dev) * 2 - 1
window = torch.hann_window(n_fft, device=dev)
desc = StftDesc(rows, L, L, n_fft, hop, n_fft, 1, 1, 0, 1, 0)
stream = P(torch.cuda.current_stream().cuda_stream)


def launch(in_ptr, out_ptr):
    if op == 'stft':
        rc = h.tac_stft_f32(P(in_ptr), P(window.data_ptr()), ctypes.byref(desc), P(out_ptr), stream)
    else:
        rc = h.tac_spectrogram_f32(P(in_ptr), P(window.data_ptr()), ctypes.byref(desc), 2.0, 0, 1.0, 1e-7, P(out_ptr), stream)
    assert rc == 0, rc


def timed(in_off, out_off, n=60):
    ins = [slab_in.data_ptr() + in_off + 4 * b * rows * L for b in range(nrot)]
    outp = slab_out.data_ptr() + out_off
    for k in range(20):
        launch(ins[k % nrot], outp)
    torch.cuda.synchronize()
    ev = []
    for k in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(ins[k % nrot], outp); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[n // 2], ts[n // 10], ts[9 * n // 10]


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    launch(slab_in.data_ptr(), slab_out.data_ptr())
    torch.cuda.synchronize()
print('%s: slab_in at 0x%x, slab_out at 0x%x' % (op, slab_in.data_ptr(), slab_out.data_ptr()))
for label, offs in (('output offset', [(0, o) for o in (0, 16, 64, 128, 256, 1024, 4096, 65536, 1 << 20, (1 << 21), (1 << 21) + 4096, 3 << 20, 1 << 24, (1 << 25) + 128)]),
                    ('input offset', [(o, 0) for o in (16, 128, 4096, 65536, 1 << 20, 1 << 21, 1 << 24)])):
    for in_off, out_off in offs:
        med, p10, p90 = timed(in_off, out_off)
        print('%-14s in +%-9d out +%-9d  median %.4f ms  p10 %.4f  p90 %.4f' % (label, in_off, out_off, med, p10, p90))
# and the same placements again in reverse order (is it the placement or the moment?)
for in_off, out_off in ((0, 1 << 24), (0, 4096), (0, 0)):
    med, p10, p90 = timed(in_off, out_off)
    print('%-14s in +%-9d out +%-9d  median %.4f ms  p10 %.4f  p90 %.4f' % ('again', in_off, out_off, med, p10, p90))
# fresh allocations, as the layer API does it: one output tensor per call from torch's caching allocator
spec = tac.Spectrogram(n_fft, hop, power=2.).to(dev) if op == 'spec' else None
xs = [torch.rand(rows, 1, L, device=dev) * 2 - 1 for _ in range(nrot)]
for rep in range(3):
    ev = []
    for k in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y = spec(xs[k % nrot]) if op == 'spec' else tac.stft(xs[k % nrot], n_fft, hop)
        y = tac.realize(y)
        b.record()
        ev.append((a, b, y.data_ptr()))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b, _ in ev)
    ptrs = sorted({p for _, _, p in ev})
    print('layer API, fresh outputs: median %.4f ms p10 %.4f p90 %.4f; distinct output addresses %s' % (ts[30], ts[6], ts[54], ['0x%x' % p for p in ptrs][:6]))
