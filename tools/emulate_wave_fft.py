#!/usr/bin/env python
"""Lane-accurate numpy emulation of the wave-level Stockham R2C FFT in csrc/fft_core.hpp.

Design aid (CPU only): checks the index algebra (pass read/write indices, twiddle indices,
LDS padding, R2C post-process pairing) and counts LDS bank conflicts of every access
pattern before any GPU time is spent.  ``python tools/emulate_wave_fft.py``.
"""
import numpy as np


def radix_plan(nc):
    """greedy 16, 16, ..., r — see fft_core.hpp for how the later passes share twiddle registers."""
    plan = []
    rem = nc
    while rem > 1:
        r = min(16, rem)
        plan.append(r)
        rem //= r
    return plan


def pad(o):
    return o + (o >> 4)


def bank_conflicts_b64(addr_complex, write):
    """addr_complex: (64,) LDS complex-element indices accessed by one wave instruction (8 B each).
    ds_read_b64: two 32-lane groups, bank=(a/4)%64;  ds_write_b64: four 16-lane groups, (a/4)%32."""
    dw = addr_complex * 2
    worst = 1
    if write:
        groups, nb = [range(i, i + 16) for i in range(0, 64, 16)], 32
    else:
        groups, nb = [range(i, i + 32) for i in range(0, 64, 32)], 64
    for g in groups:
        cnt = {}
        for lane in g:
            for d in (dw[lane], dw[lane] + 1):
                cnt.setdefault(d % nb, set()).add(d)
        worst = max(worst, max(len(v) for v in cnt.values()))
    return worst


def wave_fft(z, nc, e):
    """z: (G, nc) complex input per sub-group -> Z (G, nc).  Emulates 64 lanes."""
    lpf = nc // e
    g_per_wave = 64 // lpf
    assert z.shape == (g_per_wave, nc)
    lane = np.arange(64)
    g, t = lane // lpf, lane % lpf
    padded = nc + nc // 16
    lds = np.zeros(g_per_wave * padded + 8, dtype=np.complex128)
    gbase = g * padded
    plan = radix_plan(nc)
    s = 1
    conflicts = {}
    for pi, r in enumerate(plan):
        nb = e // r
        newvals = []
        for b in range(nb):
            j = t + b * lpf
            v = []
            for q in range(r):
                idx = j + q * (nc // r)
                if pi == 0:
                    v.append(z[g, idx])
                else:
                    a = gbase + pad(idx)
                    conflicts[('read', pi)] = max(conflicts.get(('read', pi), 1), bank_conflicts_b64(a, False))
                    v.append(lds[a])
            v = np.stack(v)                                   # (r, 64)
            if s > 1:
                k = (j % s)[None, :] * np.arange(r)[:, None]
                v = v * np.exp(-2j * np.pi * k / (s * r))
            out = np.fft.fft(v, axis=0)
            newvals.append((j, out))
        for j, out in newvals:                                # all reads precede writes (one wave, in order)
            for k in range(r):
                o = (j // s) * s * r + (j % s) + k * s
                a = gbase + pad(o)
                conflicts[('write', pi)] = max(conflicts.get(('write', pi), 1), bank_conflicts_b64(a, True))
                lds[a] = out[k]
        s *= r
    zz = np.stack([lds[gi * padded + pad(np.arange(nc))] for gi in range(g_per_wave)])
    return zz, lds, conflicts


def r2c_post(lds, nc, e):
    """Post-process from natural-order Z in LDS: X[k] for k=0..nc (emulating lane mapping)."""
    lpf = nc // e
    gpw = 64 // lpf
    padded = nc + nc // 16
    lane = np.arange(64)
    g, t = lane // lpf, lane % lpf
    n = 2 * nc
    x = np.zeros((gpw, nc + 1), dtype=np.complex128)
    npairs = max(e // 2, 1)
    conflicts = 1
    for i in range(npairs):
        k = t + i * lpf
        active = k < nc // 2 if nc >= 2 else k < 1
        kk = np.where(active, k, 0)
        a0 = g * padded + pad(kk)
        a1 = g * padded + pad((nc - kk) % nc)
        conflicts = max(conflicts, bank_conflicts_b64(a0, False), bank_conflicts_b64(a1, False))
        zk, zm = lds[a0], lds[a1]
        ev = 0.5 * (zk + np.conj(zm))
        od = -0.5j * (zk - np.conj(zm))
        tw = np.exp(-2j * np.pi * kk / n) * od
        xa, xb = ev + tw, np.conj(ev - tw)
        for ln in range(64):
            if not active[ln]:
                continue
            if kk[ln] == 0:
                z0 = lds[g[ln] * padded]
                x[g[ln], 0] = z0.real + z0.imag
                x[g[ln], nc] = z0.real - z0.imag
            else:
                x[g[ln], kk[ln]] = xa[ln]
                x[g[ln], nc - kk[ln]] = xb[ln]
    for gi in range(gpw):                                     # k = nc/2 self-paired bin
        if nc >= 2:
            x[gi, nc // 2] = np.conj(lds[gi * padded + pad(nc // 2)])
    return x, conflicts


def main():
    rng = np.random.default_rng(0)
    for nc, e in [(16, 16), (32, 16), (64, 16), (128, 16), (256, 16), (512, 16), (1024, 16), (2048, 32)]:
        lpf = nc // e
        gpw = 64 // lpf
        frames = rng.standard_normal((gpw, 2 * nc))
        z = frames[:, 0::2] + 1j * frames[:, 1::2]
        zz, lds, conf = wave_fft(z, nc, e)
        err_c = np.abs(zz - np.fft.fft(z, axis=1)).max()
        x, pconf = r2c_post(lds, nc, e)
        err_r = np.abs(x - np.fft.rfft(frames, axis=1)).max()
        print('N=%5d NC=%4d E=%2d LPF=%2d plan=%s  cfft err %.2e  rfft err %.2e  conflicts %s post %d'
              % (2 * nc, nc, e, lpf, radix_plan(nc), err_c, err_r, dict(conf), pconf))
        assert err_c < 1e-9 and err_r < 1e-9


if __name__ == '__main__':
    main()
