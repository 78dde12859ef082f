#!/usr/bin/env python
"""CPU emulation of stft_drain3_kernel's ring / span index arithmetic (csrc/stft_drain3.hpp): rows are produced into the
tightly packed LDS ring, spans are drained lane by lane exactly as the kernel computes them, and the result must equal the
output stream, with every byte of [X0, X0 + nloc * LENF) written exactly once and nothing outside it.
    python tools/r05/emulate_drain3.py"""
import random

def run(LENF, NS, nloc, X0, DW=1):
    RF = NS * LENF
    NSTMAX = (LENF + 255) // 256 + 1
    ph = X0 & 3
    ringmem = [None] * (RF + 8)
    out = {}
    wrap = lambda r: r - RF if r >= RF else r
    ring = lambda r: ringmem[ph + r]
    def produce(i):
        slot = i % NS
        for e in range(LENF):
            ringmem[ph + slot * LENF + e] = (i, e)
        if slot == 0:
            for e in range(4): ringmem[ph + RF + e] = (i, e)
    def store(x, val):
        assert x not in out, ('written twice', x)
        out[x] = val
    def drain(j):
        s1 = j % NS
        Xj = X0 + j * LENF
        lo = X0 if j == 0 else (Xj & ~255)
        hi = Xj + LENF if j == nloc - 1 else ((Xj + LENF) & ~255)
        rlo = s1 * LENF - (Xj - lo)
        if rlo < 0: rlo += RF
        lo16 = (lo + 3) & ~3
        npre = lo16 - lo
        nbody = hi - lo16
        nch, ntail = nbody >> 2, nbody & 3
        assert nch >= 1
        S = (lo16 >> 2) & 63
        NI = (nch + S + 63) >> 6
        assert NI <= NSTMAX, (NI, NSTMAX)
        for t in range(64):
            if npre and t < npre: store(lo + t, ring(wrap(rlo + t)))
            if ntail and t < ntail: store(lo16 + 4 * nch + t, ring(wrap(rlo + npre + 4 * nch + t)))
            for u in range(NI):
                c = t + 64 * u - S
                if 0 <= c < nch:
                    r = wrap(rlo + npre + 4 * c)
                    assert (ph + r) % 4 == 0, 'LDS piece not 16-byte aligned'
                    assert (lo16 + 4 * c) % 4 == 0
                    if u > 0 or S == 0 or True:
                        pass
                    for e in range(4): store(lo16 + 4 * c + e, ringmem[ph + r + e])
    # schedule: produce row j, then drain span j (needs rows j - 1 and j) — rows j - 1 ... stay until span j is read
    for j in range(nloc):
        produce(j)
        drain(j)
    for i in range(nloc):
        for e in range(LENF):
            assert out.get(X0 + i * LENF + e) == (i, e), (i, e, out.get(X0 + i * LENF + e))
    assert len(out) == nloc * LENF

random.seed(1)
n = 0
for LENF, mult in ((2050, 2), (1025, 4)):
    for NS in (mult, 2 * mult, 3 * mult):
        for nloc in (1, 2, 3, 5, 17, 40):
            for _ in range(6):
                X0 = random.randrange(0, 1 << 20) * (2 if LENF == 2050 else 1) + (1 << 22)
                run(LENF, NS, nloc, X0); n += 1
print('ok', n, 'cases')


def run_inplace(LENF, TW, nloc, X0, eager=False):
    """stft_drain3i_kernel: rows in place (area of wave i mod TW at the row's 16-byte phase), tail buffers, static dealing."""
    TB = 260
    areas = [[None] * (LENF + 3 + 64) for _ in range(TW)]
    tails = [[None] * TB for _ in range(TW)]
    NSTMAX = (LENF + 255) // 256 + 1
    cplx = LENF == 2050
    NC = 1024
    out = {}
    def store(x, val):
        assert x not in out, ('written twice', x)
        assert val is not None, ('unwritten LDS read', x)
        out[x] = val
    def produce(i):
        w = i % TW
        Xi = X0 + i * LENF
        a = Xi & 3
        ntl = (Xi + LENF) & 255
        hoff = Xi & 255
        tb_own, tb_prev = tails[w], tails[(w - 1) % TW]
        for e in range(LENF): areas[w][a + e] = (i, e)
        # tail part: elements e >= LENF - ntl, written from the lanes that hold them
        if cplx:
            for p in range(2):
                for t in range(64):
                    kk = t + 64 * p
                    if kk < (ntl >> 1):
                        c = NC - kk
                        d = (ntl >> 1) - 1 - kk
                        tb_own[2 * d] = (i, 2 * c); tb_own[2 * d + 1] = (i, 2 * c + 1)
            if i > 0 and a == 2:
                tb_prev[hoff] = (i, 0); tb_prev[hoff + 1] = (i, 1)
        else:
            npre = (4 - a) & 3
            for p in range(4):
                for t in range(64):
                    kk = t + 64 * p
                    if kk < ntl: tb_own[ntl - 1 - kk] = (i, NC - kk)
            if i > 0:
                for t in range(npre): tb_prev[hoff + t] = (i, t)
    def drain(j):
        wB = j % TW
        Xj = X0 + j * LENF
        lo = X0 if j == 0 else (Xj & ~255)
        hi = Xj + LENF if j == nloc - 1 else ((Xj + LENF) & ~255)
        lo16 = (lo + 3) & ~3
        npre = lo16 - lo
        nbody = hi - lo16
        nch, ntail = nbody >> 2, nbody & 3
        S = (lo16 >> 2) & 63
        NI = (nch + S + 63) >> 6
        assert NI <= NSTMAX
        cseam = ((((Xj + 3) & ~3) - lo16) >> 2) if j > 0 else 0
        assert cseam <= 64 - S or j == 0, 'seam beyond the first wave-store'
        area = areas[wB]
        aoff = (Xj & 3) - (Xj - lo16)
        tbuf = tails[(j - 1) % TW]
        for t in range(64):
            if npre and t < npre: store(lo + t, area[aoff + t - npre])
            if ntail and t < ntail: store(lo16 + 4 * nch + t, area[aoff + 4 * nch + t])
            for u in range(NI):
                c = t + 64 * u - S
                if 0 <= c < nch:
                    if u == 0 and c < cseam:
                        src, o = tbuf, 4 * c
                    else:
                        src, o = area, aoff + 4 * c
                        assert o >= 0 and o % 4 == 0
                    for e in range(4): store(lo16 + 4 * c + e, src[o + e])
    # the laziest legal schedule: row j is produced as soon as the protocol allows (spans j - TW and j - TW + 1 read), spans are
    # read as late as possible
    done = 0
    for j in range(nloc):
        while done <= (j - 1 if eager else j - TW + 1):
            drain(done); done += 1
        produce(j)
    while done < nloc:
        drain(done); done += 1
    for i in range(nloc):
        for e in range(LENF):
            assert out.get(X0 + i * LENF + e) == (i, e), (i, e, out.get(X0 + i * LENF + e))
    assert len(out) == nloc * LENF

n = 0
for LENF in (2050, 1025):
    for TW in (2, 9, 12, 15):
        for nloc in (1, 2, 3, 5, 17, 40):
            for _ in range(6):
                X0 = random.randrange(0, 1 << 20) * (4 if LENF == 2050 else 4) + (1 << 22) + (random.randrange(0, 2000) * LENF)
                run_inplace(LENF, TW, nloc, X0); run_inplace(LENF, TW, nloc, X0, eager=True); n += 2
print('ok in-place', n, 'cases')
