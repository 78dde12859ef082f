// exact_math.hpp — float32 log1p with the reference CPU path's exact roundings.
// Plain C++ when compiled by g++ (tools/check_log1p_replica.py builds it that way to compare against torch CPU),
// __host__ __device__ under hipcc.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define TAC_HD __host__ __device__
#else
#define TAC_HD
#endif

namespace tac {

// log1p exactly as the reference's CPU path evaluates it.  functional.py:333 calls torch.log1p on float32, which
// torch's CPU backend vectorises with Sleef's 1.0-ULP log1pf (FMA build): an exponent split e = ilogb((1+d)/0.75),
// m = d*2^-e + (2^-e - 1), x = m/(2+m) in float-float arithmetic, an odd polynomial in x, and e*ln2 added as a
// float-float constant.  Reproducing those roundings one by one (no FMA contraction beyond the explicit fmaf's,
// correctly rounded division) makes every mu-law code bit-identical to the reference for any n_quantize and any
// input range; tools/check_log1p_replica.py holds the exhaustive host-side comparison (0 mismatches over every
// float32 in [0, 256) and strided up to 2^23), tests/test_gpu_parity.py the device-side one against tests/golden.
struct ff2 { float hi, lo; };

TAC_HD inline float exact_log1pf(float d) {
#pragma clang fp contract(off)
    if (!(d <= 1e38f)) return d != d ? d : logf(d);               // beyond the core's exponent range (|x| > 1e35/mu)
    if (d == 0.0f) return d;
    const float dp1 = d + 1.0f;
    const float sc = dp1 * (1.0f / 0.75f);
    unsigned ub;
    __builtin_memcpy(&ub, &sc, 4);
    const int e = (int)((ub >> 23) & 0xffu) - 0x7f;
    ub = (unsigned)(0x7f - e) << 23;
    float t;
    __builtin_memcpy(&t, &ub, 4);
    const float m = __builtin_fmaf(d, t, t - 1.0f);
    const float ef = (float)e;
    // e * ln2 (float-float constant times float)
    ff2 s;
    s.hi = 0.69314718246459960938f * ef;
    s.lo = __builtin_fmaf(-1.904654323148236017e-09f, ef, __builtin_fmaf(0.69314718246459960938f, ef, -s.hi));
    // x = m / (2 + m)
    ff2 den;
    den.hi = 2.0f + m;
    den.lo = 2.0f - den.hi + m;
    const float rcp = 1.0f / den.hi;
    ff2 x;
    x.hi = m * rcp;
    const float u = __builtin_fmaf(rcp, m, -x.hi);
    x.lo = __builtin_fmaf(-den.lo, rcp, __builtin_fmaf(-den.hi, rcp, 1.0f));
    x.lo = __builtin_fmaf(x.hi, x.lo, __builtin_fmaf(0.0f, rcp, u));
    const float x2 = x.hi * x.hi;
    float p = 0.3027294874e+0f;
    p = __builtin_fmaf(p, x2, 0.3996108174e+0f);
    p = __builtin_fmaf(p, x2, 0.6666694880e+0f);
    // s += 2x ; s += x^3 * p
    const float ax = x.hi * 2.0f, ay = x.lo * 2.0f;
    ff2 a;
    a.hi = s.hi + ax;
    a.lo = s.hi - a.hi + ax + s.lo + ay;
    const float c = x2 * x.hi * p;
    ff2 b;
    b.hi = a.hi + c;
    b.lo = a.hi - b.hi + c + a.lo;
    return b.hi + b.lo;
}

}  // namespace tac
