"""Lane-accurate CPU emulation of the matrix-pipe form of the 1024-point complex transform (csrc/fft_mfma.hpp, round 6).

The transform of one fft_length-2048 frame (packed as 1024 complex values z[m] = x[2m] + i x[2m+1]) is two chained
32 x 32 complex DFT products with a twiddle in between:

    n = 32 n1 + n2,  k = k1 + 32 k2
    Y[n2][k1]  = sum_n1 z[32 n1 + n2] W32^(n1 k1)                (step 1: A = data, B = constant)
    Y'[n2][k1] = Y[n2][k1] W1024^(n2 k1)                         (VALU)
    Z[k1 + 32 k2] = sum_n2 W32^(k2 n2) Y'[n2][k1]                (step 2: A = constant, B = data)

evaluated on v_mfma_f32_32x32x16_f16 with every float32 operand split into an fp16 (hi, lo) pair and the three products
hi.hi + hi.lo + lo.hi accumulated in float32.  This script checks (1) the index algebra against numpy.fft, including the
register / lane placement of the MFMA operands (the D registers of step 1 ARE the B operand of step 2 once the K order of the
constant matrix follows them), and (2) the accuracy of the split against a float64 transform.

    python tools/emulate_mfma_fft.py
"""
import numpy as np

N = 1024


def perm(c, h, j):
    """K slot (mfma c, lane half h, element j of the lane's 8) -> summation index; dictated by the D layout of
    v_mfma_f32_32x32x*: register r = 8 c + j of half h holds row 8 (r >> 2) + 4 h + (r & 3)."""
    r = 8 * c + j
    return 8 * (r >> 2) + 4 * h + (r & 3)


def split16(x, rtz=False):
    """float32 -> (hi, lo) fp16 pair with hi + lo ~ x"""
    x = x.astype(np.float32)
    if rtz:
        bits = x.view(np.uint32) & np.uint32(0xFFFFE000)
        hi = bits.view(np.float32).astype(np.float16)
    else:
        hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def mfma_32x32x16(a, b, acc):
    """a: (32, 16) fp16 rows x K, b: (16, 32) fp16 K x cols, acc (32, 32) float32"""
    return (acc.astype(np.float64) + a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)


def frame_scale_exp(v):
    """power-of-two exponent e with max |v| * 2^-e in [2^7, 2^8)"""
    m = float(np.abs(v).max())
    if m == 0.0 or not np.isfinite(m):
        return 0
    return int(np.floor(np.log2(m))) - 7


def transform(z, rtz=False):
    """z: (1024,) complex64 (already windowed).  Returns Z (1024,) complex64 via the emulated MFMA data flow."""
    zr, zi = z.real.astype(np.float32), z.imag.astype(np.float32)
    e = frame_scale_exp(np.concatenate([zr, zi]))
    s = np.float32(2.0 ** -e)
    zr, zi = zr * s, zi * s
    w32 = np.exp(-2j * np.pi * np.outer(np.arange(32), np.arange(32)) / 32)
    wr_h, wr_l = split16(w32.real.astype(np.float32))
    wi_h, wi_l = split16(w32.imag.astype(np.float32))

    # K order shared by both steps
    korder = np.array([[perm(c, h, j) for h in range(2) for j in range(8)] for c in range(2)])        # [c][16 slots]

    # ---- step 1: A[row n2][K = n1] = z[32 n1 + n2]; B[K = n1][col k1] = W32[n1][k1]; D[n2][k1]
    xr = zr.reshape(32, 32).T          # [n2][n1]
    xi = zi.reshape(32, 32).T
    xr_h, xr_l = split16(xr, rtz)
    xi_h, xi_l = split16(xi, rtz)
    dr = np.zeros((32, 32), np.float32)
    di = np.zeros((32, 32), np.float32)
    for c in range(2):
        ks = korder[c]
        for (ah, al, bh, bl, sign, into) in (
            (xr_h, xr_l, wr_h, wr_l, 1, 'r'), (xi_h, xi_l, wi_h, wi_l, -1, 'r'),
            (xr_h, xr_l, wi_h, wi_l, 1, 'i'), (xi_h, xi_l, wr_h, wr_l, 1, 'i')):
            for (a, b) in ((ah, bh), (ah, bl), (al, bh)):
                aa = a[:, ks] if sign > 0 else -a[:, ks]
                if into == 'r':
                    dr = mfma_32x32x16(aa, b[ks, :], dr)
                else:
                    di = mfma_32x32x16(aa, b[ks, :], di)
    # ---- twiddle W1024^(n2 k1) on D[n2][k1] (float32 VALU)
    tw = np.exp(-2j * np.pi * np.outer(np.arange(32), np.arange(32)) / 1024)
    twr, twi = tw.real.astype(np.float32), tw.imag.astype(np.float32)
    yr = dr * twr - di * twi
    yi = dr * twi + di * twr
    # ---- step 2: A[row k2][K = n2] = W32[k2][n2]; B[K = n2][col k1] = Y'[n2][k1]; D[k2][k1] = Z[k1 + 32 k2]
    yr_h, yr_l = split16(yr, rtz)
    yi_h, yi_l = split16(yi, rtz)
    dr = np.zeros((32, 32), np.float32)
    di = np.zeros((32, 32), np.float32)
    for c in range(2):
        ks = korder[c]
        for (ah, al, bh, bl, sign, into) in (
            (wr_h, wr_l, yr_h, yr_l, 1, 'r'), (wi_h, wi_l, yi_h, yi_l, -1, 'r'),
            (wi_h, wi_l, yr_h, yr_l, 1, 'i'), (wr_h, wr_l, yi_h, yi_l, 1, 'i')):
            for (a, b) in ((ah, bh), (ah, bl), (al, bh)):
                bb = b[ks, :] if sign > 0 else -b[ks, :]
                if into == 'r':
                    dr = mfma_32x32x16(a[:, ks], bb, dr)
                else:
                    di = mfma_32x32x16(a[:, ks], bb, di)
    Z = (dr.astype(np.float64) + 1j * di.astype(np.float64)).reshape(-1) * 2.0 ** e       # [k2][k1] -> k = k1 + 32 k2
    return Z


def lane_register_checks():
    """The placement the kernel relies on: step-1 D register r of lane (k1, h) holds n2 = 8 (r >> 2) + 4 h + (r & 3), which is
    exactly the K slot (c = r >> 3, h, j = r & 7) of step 2's B operand; lower / upper halves of the spectrum by register."""
    for h in range(2):
        for r in range(16):
            assert perm(r >> 3, h, r & 7) == 8 * (r >> 2) + 4 * h + (r & 3)
    # every K index appears once per (c) pair of MFMAs
    for c in range(2):
        ks = sorted(perm(c, h, j) for h in range(2) for j in range(8))
        assert ks == list(range(16 * c, 16 * c + 16))
    # registers 0..7 hold k2 < 16 (bins k < 512), registers 8..15 the upper half; the partner of bin k = k1 + 32 k2 (k1 > 0) is
    # lane (32 - k1, 1 - h), register 15 - r
    for h in range(2):
        for r in range(16):
            k2 = 8 * (r >> 2) + 4 * h + (r & 3)
            assert (k2 < 16) == (r < 8)
            r2, h2 = 15 - r, 1 - h
            assert 8 * (r2 >> 2) + 4 * h2 + (r2 & 3) == 31 - k2


def main():
    lane_register_checks()
    rng = np.random.default_rng(0)
    win = np.hanning(2049)[:2048].astype(np.float32)
    worst = {}
    for name, gen in (('normal', lambda: rng.standard_normal(2048)),
                      ('uniform', lambda: rng.uniform(-1, 1, 2048)),
                      ('tone+noise', lambda: np.sin(0.05 * np.arange(2048)) + 1e-4 * rng.standard_normal(2048)),
                      ('pcm16', lambda: np.round(rng.uniform(-32768, 32767, 2048))),
                      ('tiny', lambda: 1e-20 * rng.standard_normal(2048)),
                      ('impulse', lambda: np.eye(1, 2048, 777)[0] * 3.0)):
        for rtz in (False, True):
            errs = []
            for _ in range(4):
                x = gen().astype(np.float32) * win
                z = (x[0::2] + 1j * x[1::2]).astype(np.complex64)
                want = np.fft.fft(z.astype(np.complex128))
                got = transform(z, rtz)
                f32 = np.fft.fft(z).astype(np.complex64)       # numpy's own single-precision-ish path for scale
                errs.append(np.abs(got - want).max() / np.abs(want).max())
            worst[(name, rtz)] = max(errs)
            print('%-12s %s  max |err| / max |Z| = %.3g' % (name, 'rtz' if rtz else 'rne', max(errs)))
    assert max(worst.values()) < 1e-6
    print('ok')


if __name__ == '__main__':
    main()
