#!/usr/bin/env python
"""Same-process A/B of library builds for the SURVEY 8(f) rows (phase_vocoder, hpss): like ab_inproc.py, through the C ABI.
    python tools/r04/ab_other.py pv[:rate] | hpss[:k | :kfxkt] name=path [name=path ...]
256 x 1025 x 313 frame-major complex spectrogram (cfg-2's STFT output); prints median / p10 / p90 per build and the median of
per-round differences to the first build; the first launch of every build is compared with the first build's."""
import ctypes, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd import _hip

op = sys.argv[1]
builds = [a.split('=', 1) for a in sys.argv[2:]]
P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
libs = []
for name, path in builds:
    h = ctypes.CDLL(os.path.abspath(path))
    h.tac_phase_vocoder_f32.argtypes = [P, I64, I32, I64, I64, I64, I64, P, P, P, P, I64, P, P]
    h.tac_hpss_f32.argtypes = [P, I64, I32, I32, I64, I64, I64, I32, I32, ctypes.c_float, ctypes.c_int, P, P, P, P, P]
    libs.append((name, h))
dev = torch.device('cuda', 0)
N = int(os.environ.get('TAC_AB_N', '200'))
nrot = int(os.environ.get('TAC_ROTATE', '2'))
rows, F, T = 256, 1025, 313
stream = P(torch.cuda.current_stream().cuda_stream)
if op.startswith('pv'):
    rate = float(op.split(':')[1]) if ':' in op else 1.3
    zs = [torch.randn(rows, T, F, 2, device=dev) for _ in range(nrot)]
    adv = torch.linspace(0, math.pi * 512, F, device=dev)
    idx0, idx1, alpha = _hip._phase_vocoder_grid(T, rate, dev, torch.float32)
    n_out = idx0.numel()
    out = torch.empty(rows, n_out, F, 2, device=dev)
    nbytes = zs[0].numel() * 4 + out.numel() * 4

    def launch(name, h, z):
        rc = h.tac_phase_vocoder_f32(P(z.data_ptr()), rows, F, T, T * F * 2, 2, F * 2, P(adv.data_ptr()), P(idx0.data_ptr()),
                                     P(idx1.data_ptr()), P(alpha.data_ptr()), n_out, P(out.data_ptr()), stream)
        assert rc == 0, (name, rc)
elif op.startswith('hpss'):
    ksz = op.split(':')[1] if ':' in op else '31'
    kf, kt = (int(v) for v in (ksz.split('x') if 'x' in ksz else (ksz, ksz)))        # hpss:31 or hpss:5x9 (freq x time)
    zs = [torch.rand(rows, T, F, device=dev) ** 2 for _ in range(nrot)]          # frame-major |X|^2, as the STFT kernels return it
    outs = [torch.empty(rows, T, F, device=dev) for _ in range(4)]
    out = outs[0]
    nbytes = zs[0].numel() * 20

    def launch(name, h, z):
        rc = h.tac_hpss_f32(P(z.data_ptr()), rows, F, T, T * F, 1, F, kf, kt, 2.0, 0, P(outs[0].data_ptr()), P(outs[1].data_ptr()),
                            P(outs[2].data_ptr()), P(outs[3].data_ptr()), stream)
        assert rc == 0, (name, rc)
else:
    raise SystemExit('unknown op ' + op)

ref = None
for name, h in libs:
    launch(name, h, zs[0]); torch.cuda.synchronize()
    cur = out.clone()
    if ref is None: ref = cur
    else: print('check %-10s max |diff| vs %s: %.3g (scale %.3g)' % (name, libs[0][0], (cur - ref).abs().max().item(), ref.abs().max().item()))
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 1.0:
    for name, h in libs:
        launch(name, h, zs[k % nrot]); k += 1
    torch.cuda.synchronize()
ev = {name: [] for name, _ in libs}
for r in range(N):
    order = libs if r % 2 == 0 else libs[::-1]
    for name, h in order:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(name, h, zs[k % nrot]); b.record(); k += 1
        ev[name].append((a, b))
    if nrot % len(libs) == 0:
        k += 1
torch.cuda.synchronize()
ts = {name: [a.elapsed_time(b) for a, b in v] for name, v in ev.items()}
q = lambda v, f: sorted(v)[int(f * (len(v) - 1))]
base = libs[0][0]
for name, _ in libs:
    v = ts[name]
    diff = sorted(x - y for x, y in zip(v, ts[base]))
    print('%-7s %-10s median %.4f ms (%.2f TB/s)  p10 %.4f  p90 %.4f | vs %s: %+.4f ms (%+.2f %%)'
          % (op, name, q(v, .5), nbytes / q(v, .5) / 1e9, q(v, .1), q(v, .9), base, q(diff, .5), 100 * q(diff, .5) / q(ts[base], .5)))
