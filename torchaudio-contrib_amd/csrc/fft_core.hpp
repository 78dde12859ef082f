// fft_core.hpp — wave-level Stockham R2C FFT for gfx950 (wave64).
//
// One N-point real frame = one NC=N/2-point complex FFT of the packed (even, odd) samples plus
// an R2C split.  A frame is owned by LPF = NC/E lanes of ONE wavefront, E complex elements per
// lane (E=16 up to N=2048, 32 for N=4096); G = 64/LPF frames share a wave.  Because a frame never
// leaves its wave, the inter-pass exchanges through LDS need no s_barrier: LDS executes one wave's
// DS instructions in issue order, so a wave-level compiler fence is all the ordering required.
//
// Passes are radix-16 (in registers, 4x4) with a radix-{2,4,8,16} tail: N=2048 is 16·16·4, i.e.
// two LDS exchanges + one for the R2C pairing.  LDS image of a frame is NC complex padded by one
// element per 16 (index o -> o + o/16): that makes the stride-R writes of the first pass
// bank-conflict free (checked lane-accurately in tools/emulate_wave_fft.py).
//
// Inter-pass twiddles, the window and the R2C twiddles depend only on the lane, not on the frame,
// so persistent kernels load them ONCE into registers and reuse them for every frame.
#pragma once
#include <hip/hip_runtime.h>

namespace tac {

// Complex value = one 64-bit register pair.  The arithmetic below is written for gfx950's PACKED f32 VALU ops
// (v_pk_add/mul/fma_f32: two lanes of a register pair per instruction, with per-operand half selection and sign
// modifiers): a complex add is one instruction, a complex multiply two, a radix-4 butterfly eight.  A packed op
// occupies the SIMD as long as two scalar ones, but it takes ONE issue slot of the wave — and these kernels run
// two 256-register waves per SIMD, each limited to about one VALU issue per four cycles (measured,
// tools/ubench/valu_rate.hip), so per-wave issue slots, not SIMD cycles, are what the FFT is short of.
// The rotations / conjugations are folded into the consuming add via op_sel / neg modifiers, which hipcc does not
// derive from shuffles by itself (it emits v_mov + v_xor), hence the one-line asm helpers.
typedef float cf __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ cf mkc(float x, float y) {
    cf r;
    r.x = x;
    r.y = y;
    return r;
}

// Complex arithmetic on register pairs with the packed-f32 VALU ops (v_pk_*_f32 and their op_sel / neg modifiers).
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
// a + (-i)·b and a - (-i)·b
__device__ __forceinline__ cf cadd_rot(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf csub_rot(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b) and a - conj(b)
__device__ __forceinline__ cf cadd_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf csub_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// conj(a - b)
__device__ __forceinline__ cf csub_then_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a · w (w a register twiddle): (a.x·w) + a.y·(-w.y, w.x)
__device__ __forceinline__ cf cmul(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// two independent products a·wa, b·wb with their instruction pairs interleaved: gfx950 needs a wait state between a
// packed op and a dependent packed op that follows it directly, and hipcc fills it with an s_nop (an issue slot) when
// the two asm statements of one cmul stay adjacent
__device__ __forceinline__ void cmul_x2(cf& a, cf wa, cf& b, cf wb) {
    cf t1, t2, r1, r2;
    asm("v_pk_mul_f32 %0, %4, %5 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %6, %7 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %4, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %6, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t1), "=&v"(t2), "=&v"(r1), "=&v"(r2) : "v"(a), "v"(wa), "v"(b), "v"(wb));
    a = r1;
    b = r2;
}
// w · (-i·d): the R2C split's twiddled odd part, d = zk - conj(zm)
__device__ __forceinline__ cf cmul_rot(cf w, cf d) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(d), "v"(w));                    // d.y·w
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(d), "v"(w), "v"(t));   // + d.x·(w.y, -w.x)
    return r;
}
// (|a + b|^2, |a - b|^2): the real parts of both sums in one register pair, the imaginary parts in another, so the
// two squared magnitudes cost one packed multiply and one packed FMA
__device__ __forceinline__ cf power_pair(cf a, cf b) {
    cf re, im;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(re) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(im) : "v"(a), "v"(b));
    return __builtin_elementwise_fma(im, im, re * re);
}
// (|2X[k]|^2, |2X[NC-k]|^2) of TWO R2C pairs (zk, zm, wk) at once, the sixteen packed instructions interleaved so that no
// instruction directly follows its producer (see cmul_x2):  ev = zk + conj(zm), d = zk - conj(zm), tw = wk·(-i·d),
// power_pair(ev, tw)
__device__ __forceinline__ void r2c_power_pair_x2(cf zk1, cf zm1, cf w1, cf zk2, cf zm2, cf w2, cf& p1, cf& p2) {
    cf ev1, ev2, d1, d2, t1, t2, tw1, tw2, re1, re2, im1, im2, q1, q2;
    asm("v_pk_add_f32 %0, %16, %17 neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %19, %20 neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %16, %17 neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %3, %19, %20 neg_lo:[0,1]\n\t"
        "v_pk_mul_f32 %4, %2, %18 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %5, %3, %21 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_fma_f32 %6, %2, %18, %4 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %7, %3, %21, %5 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]\n\t"
        "v_pk_add_f32 %8, %0, %6 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %9, %1, %7 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %10, %0, %6 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %11, %1, %7 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %12, %8, %8\n\t"
        "v_pk_mul_f32 %13, %9, %9\n\t"
        "v_pk_fma_f32 %14, %10, %10, %12\n\t"
        "v_pk_fma_f32 %15, %11, %11, %13"
        : "=&v"(ev1), "=&v"(ev2), "=&v"(d1), "=&v"(d2), "=&v"(t1), "=&v"(t2), "=&v"(tw1), "=&v"(tw2), "=&v"(re1), "=&v"(re2),
          "=&v"(im1), "=&v"(im2), "=&v"(q1), "=&v"(q2), "=&v"(p1), "=&v"(p2)
        : "v"(zk1), "v"(zm1), "v"(w1), "v"(zk2), "v"(zm2), "v"(w2));
}
// a · (C + iS) with compile-time C, S: the constant pairs live in SGPRs
__device__ __forceinline__ cf cmulc(cf a, float C, float S) { return mkc(a.x, a.x) * mkc(C, S) + mkc(a.y, a.y) * mkc(-S, C); }
__device__ __forceinline__ cf cscale(cf a, float s) { return a * mkc(s, s); }
__device__ __forceinline__ cf cmul_elem(cf a, cf b) { return a * b; }        // lane-wise product (window)
// a * (-i)
__device__ __forceinline__ cf mul_neg_i(cf a) { return mkc(a.y, -a.x); }
__device__ __forceinline__ float cnorm2(cf a) { return a.x * a.x + a.y * a.y; }

// Compiler-only ordering point for same-wave LDS traffic (no instruction emitted).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Phase stamps for timing builds (tools/ablation/lab_knobs_r06.patch restores them): NoStamp compiles to nothing.
struct NoStamp {
    __device__ __forceinline__ void mark(int) {}
};
struct CycleStamp {
    long long last;
    float acc[12];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
        last = clock64();
    }
    __device__ __forceinline__ void mark(int i) {
        __builtin_amdgcn_sched_barrier(0);
        const long long now = clock64();
        acc[i] += (float)(now - last);
        last = now;
        __builtin_amdgcn_sched_barrier(0);
    }
};

__device__ __forceinline__ int lds_pad(int o) { return o + (o >> 4); }
// padded offset of a compile-time multiple of 16: pad(a + c) == pad(a) + lds_pad_c(c)
__device__ __forceinline__ constexpr int lds_pad_c(int c) { return c + c / 16; }

// ---------------------------------------------------------------- in-register DFTs (forward, natural order)
__device__ __forceinline__ void dft2(cf& a, cf& b) {
    cf t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

__device__ __forceinline__ void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf s02 = cadd(a0, a2), d02 = csub(a0, a2);
    cf s13 = cadd(a1, a3), d13 = csub(a1, a3);
    a0 = cadd(s02, s13);
    a1 = cadd_rot(d02, d13);
    a2 = csub(s02, s13);
    a3 = csub_rot(d02, d13);
}
// the same butterfly on lane-wise WINDOWED inputs x_i ∘ w_i: the window products are folded into the first stage
// (one multiply + two FMAs per pair instead of two multiplies, an add and a subtract)
__device__ __forceinline__ void dft4_windowed(cf& a0, cf& a1, cf& a2, cf& a3, cf w0, cf w1, cf w2, cf w3) {
    const cf t0 = cmul_elem(a0, w0), t1 = cmul_elem(a1, w1);
    const cf nw2 = mkc(-w2.x, -w2.y), nw3 = mkc(-w3.x, -w3.y);
    cf s02 = __builtin_elementwise_fma(a2, w2, t0), d02 = __builtin_elementwise_fma(a2, nw2, t0);
    cf s13 = __builtin_elementwise_fma(a3, w3, t1), d13 = __builtin_elementwise_fma(a3, nw3, t1);
    a0 = cadd(s02, s13);
    a1 = cadd_rot(d02, d13);
    a2 = csub(s02, s13);
    a3 = csub_rot(d02, d13);
}
// the same butterfly when input a2 still carries a pending factor (-i): a2 <- (-i)·a2 folded into the first stage
__device__ __forceinline__ void dft4_rot2(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf s02 = cadd_rot(a0, a2), d02 = csub_rot(a0, a2);
    cf s13 = cadd(a1, a3), d13 = csub(a1, a3);
    a0 = cadd(s02, s13);
    a1 = cadd_rot(d02, d13);
    a2 = csub(s02, s13);
    a3 = csub_rot(d02, d13);
}

#define TAC_SQRT_HALF 0.70710678118654752440f
#define TAC_COS_PI_8 0.92387953251128675613f
#define TAC_SIN_PI_8 0.38268343236508977173f

// multiply by W16^M = exp(-2*pi*i*M/16), M compile-time
template <int M>
__device__ __forceinline__ cf mul_w16(cf v) {
    if constexpr (M == 0) return v;
    else if constexpr (M == 4) return mul_neg_i(v);
    else if constexpr (M == 2) return cscale(cadd_rot(v, v), TAC_SQRT_HALF);        // (x+y, y-x)/sqrt2
    else if constexpr (M == 6) return cscale(csub_rot(v, v), -TAC_SQRT_HALF);       // (y-x, -(x+y))/sqrt2
    else if constexpr (M == 1) return cmulc(v, TAC_COS_PI_8, -TAC_SIN_PI_8);
    else if constexpr (M == 3) return cmulc(v, TAC_SIN_PI_8, -TAC_COS_PI_8);
    else if constexpr (M == 9) return cmulc(v, -TAC_COS_PI_8, TAC_SIN_PI_8);
    else return v;
}

// multiply by W32^m = exp(-2*pi*i*m/32).  `m` is a loop index of fully unrolled loops, so after unrolling
// every call sees a constant: the table lookups fold to literals and the quarter/eighth-turn cases to
// add/sub/swap forms.
__device__ __forceinline__ cf mul_w32(cf v, int m) {
    constexpr float C[32] = {1.0000000000e+00f, 9.8078528040e-01f, 9.2387953251e-01f, 8.3146961230e-01f, 7.0710678119e-01f, 5.5557023302e-01f, 3.8268343237e-01f, 1.9509032202e-01f, 6.1232339957e-17f, -1.9509032202e-01f, -3.8268343237e-01f, -5.5557023302e-01f, -7.0710678119e-01f, -8.3146961230e-01f, -9.2387953251e-01f, -9.8078528040e-01f, -1.0000000000e+00f, -9.8078528040e-01f, -9.2387953251e-01f, -8.3146961230e-01f, -7.0710678119e-01f, -5.5557023302e-01f, -3.8268343237e-01f, -1.9509032202e-01f, -1.8369701987e-16f, 1.9509032202e-01f, 3.8268343237e-01f, 5.5557023302e-01f, 7.0710678119e-01f, 8.3146961230e-01f, 9.2387953251e-01f, 9.8078528040e-01f};
    constexpr float S[32] = {0.0000000000e+00f, -1.9509032202e-01f, -3.8268343237e-01f, -5.5557023302e-01f, -7.0710678119e-01f, -8.3146961230e-01f, -9.2387953251e-01f, -9.8078528040e-01f, -1.0000000000e+00f, -9.8078528040e-01f, -9.2387953251e-01f, -8.3146961230e-01f, -7.0710678119e-01f, -5.5557023302e-01f, -3.8268343237e-01f, -1.9509032202e-01f, -1.2246467991e-16f, 1.9509032202e-01f, 3.8268343237e-01f, 5.5557023302e-01f, 7.0710678119e-01f, 8.3146961230e-01f, 9.2387953251e-01f, 9.8078528040e-01f, 1.0000000000e+00f, 9.8078528040e-01f, 9.2387953251e-01f, 8.3146961230e-01f, 7.0710678119e-01f, 5.5557023302e-01f, 3.8268343237e-01f, 1.9509032202e-01f};
    m &= 31;
    if (m == 0) return v;
    if (m == 8) return mul_neg_i(v);
    if (m == 16) return mkc(-v.x, -v.y);
    if (m == 24) return mkc(-v.y, v.x);
    if (m == 4) return cscale(cadd_rot(v, v), TAC_SQRT_HALF);
    if (m == 12) return cscale(csub_rot(v, v), -TAC_SQRT_HALF);
    return cmulc(v, C[m], S[m]);
}

template <int R>
struct Dft;

template <>
struct Dft<2> {
    __device__ static __forceinline__ void run(cf* v) { dft2(v[0], v[1]); }
};
template <>
struct Dft<4> {
    __device__ static __forceinline__ void run(cf* v) { dft4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Dft<8> {
    __device__ static __forceinline__ void run(cf* v) {
        cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        dft4(e0, e1, e2, e3);
        dft4(o0, o1, o2, o3);
        o1 = mul_w16<2>(o1);   // W8^1
        o2 = mul_w16<4>(o2);   // W8^2
        o3 = mul_w16<6>(o3);   // W8^3
        v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
        v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
        v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
        v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
    }
};
template <>
struct Dft<16> {
    // n = c + 4d, k = r + 4k':  X[r+4k'] = sum_c W4^{ck'} W16^{cr} (sum_d x[c+4d] W4^{dr})
    __device__ static __forceinline__ void run(cf* v) { run_impl<false>(v, nullptr); }
    // v holds raw samples, win[e] the lane's window pair of element e: transforms v ∘ win
    __device__ static __forceinline__ void run_windowed(cf* v, const cf* win) { run_impl<true>(v, win); }
    template <bool WINDOWED>
    __device__ static __forceinline__ void run_impl(cf* v, const cf* win) {
        cf a[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[c][0] = v[c]; a[c][1] = v[c + 4]; a[c][2] = v[c + 8]; a[c][3] = v[c + 12];
            if constexpr (WINDOWED) dft4_windowed(a[c][0], a[c][1], a[c][2], a[c][3], win[c], win[c + 4], win[c + 8], win[c + 12]);
            else dft4(a[c][0], a[c][1], a[c][2], a[c][3]);
        }
        a[1][1] = mul_w16<1>(a[1][1]); a[1][2] = mul_w16<2>(a[1][2]); a[1][3] = mul_w16<3>(a[1][3]);
        a[2][1] = mul_w16<2>(a[2][1]); /* a[2][2]·W16^4 = -i: folded into column 2's butterfly */ a[2][3] = mul_w16<6>(a[2][3]);
        a[3][1] = mul_w16<3>(a[3][1]); a[3][2] = mul_w16<6>(a[3][2]); a[3][3] = mul_w16<9>(a[3][3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cf b0 = a[0][r], b1 = a[1][r], b2 = a[2][r], b3 = a[3][r];
            if (r == 2) dft4_rot2(b0, b1, b2, b3);
            else dft4(b0, b1, b2, b3);
            v[r] = b0; v[r + 4] = b1; v[r + 8] = b2; v[r + 12] = b3;
        }
    }
};

// ---------------------------------------------------------------- plan algebra (compile time)
constexpr int num_passes(int nc) {
    int n = 0, rem = nc;
    while (rem > 1) { rem /= (rem >= 16 ? 16 : rem); ++n; }
    return n;
}
// Radix order: greedy 16, 16, ..., r (N=2048: 16·16·4).  Twiddle registers are what this design is short of
// (everything lane-dependent is hoisted for the kernel's lifetime), so both later passes are arranged to
// need only R-1 lane-dependent factors each:
//  * a middle pass whose butterflies b = 0..E/R-1 use j = t + b*LPF with LPF a multiple of the stride S:
//    they all see the same j mod S and share one set;
//  * the LAST pass, where j mod S = j = t + b*LPF: W_NC^{(t + b*LPF) q} = W_NC^{t q} · W_E^{b q}, i.e. one
//    lane-dependent set W_NC^{t q} times COMPILE-TIME constants W_E^{b q}.
// N=2048 needs 15 + 3 = 18 complex twiddle registers instead of 54 (plain) — at the price of 9 constant
// multiplies per frame.
constexpr int radix_at(int nc, int pass) {
    int rem = nc;
    for (int p = 0; p < pass; ++p) rem /= (rem >= 16 ? 16 : rem);
    return rem >= 16 ? 16 : rem;
}
constexpr int stride_at(int nc, int pass) {
    int s = 1;
    for (int p = 0; p < pass; ++p) s *= radix_at(nc, p);
    return s;
}
constexpr bool pass_is_last(int nc, int pass) { return pass + 1 == num_passes(nc); }
constexpr bool pass_shares_twiddles(int nc, int e, int pass) {
    return pass_is_last(nc, pass) || ((nc / e) % stride_at(nc, pass)) == 0;
}
constexpr int pass_twiddles(int nc, int e, int pass) {
    int r = radix_at(nc, pass);
    return (pass_shares_twiddles(nc, e, pass) ? 1 : e / r) * (r - 1);
}
constexpr int twiddles_before(int nc, int e, int pass) {   // hoisted-twiddle registers used by passes < pass
    int n = 0;
    for (int p = 1; p < pass; ++p) n += pass_twiddles(nc, e, p);
    return n;
}

template <int NC_, int E_>
struct WaveFft {
    static constexpr int NC = NC_;
    static constexpr int E = E_;
    static constexpr int N = 2 * NC;
    static constexpr int LPF = NC / E;            // lanes per frame
    static constexpr int G = 64 / LPF;            // frames per wave
    static constexpr int PADDED = NC + NC / 16 + 1;   // LDS complex slots per frame
    static constexpr int NPASS = num_passes(NC);
    static constexpr int NTW = twiddles_before(NC, E, NPASS) > 0 ? twiddles_before(NC, E, NPASS) : 1;
    static constexpr int NPAIR = E / 2;           // R2C pairs per lane
    static_assert(LPF >= 1 && LPF <= 64 && LPF * E == NC, "bad (NC, E)");

    // W_NC table -> registers.  table[i] = exp(-2*pi*i*i/NC), i < NC.
    __device__ static __forceinline__ void load_twiddles(cf* tw, const cf* __restrict__ table, int t) {
        load_tw_pass<1>(tw, table, t);
    }
    template <int P>
    __device__ static __forceinline__ void load_tw_pass(cf* tw, const cf* __restrict__ table, int t) {
        if constexpr (P < NPASS) {
            constexpr int R = radix_at(NC, P), S = stride_at(NC, P), OFF = twiddles_before(NC, E, P);
            constexpr int NB = pass_shares_twiddles(NC, E, P) ? 1 : E / R;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                int j = t + b * LPF;
                int js = j & (S - 1);
#pragma unroll
                for (int q = 1; q < R; ++q) tw[OFF + b * (R - 1) + q - 1] = table[js * q * (NC / (S * R))];
            }
            load_tw_pass<P + 1>(tw, table, t);
        }
    }

    // W_E^{b q} factor of the last pass (identity elsewhere); M32 = b*q*32/E
    template <int P>
    __device__ static __forceinline__ cf last_pass_const(cf x, int m32) {
        if constexpr (pass_is_last(NC, P) && P > 0) return mul_w32(x, m32);
        else return x;
    }

    // v[f]: E registers per frame in first-pass order v[f][b*R0 + q] = z_f[(t + b*LPF) + q*NC/R0].
    // NF independent frames are advanced pass by pass together so that one frame's LDS round trip
    // hides behind the other's butterflies (ILP instead of occupancy: the kernels run 2 waves/SIMD).
    // On return frame f's spectrum Z[0..NC) sits in natural order at lds[f][lds_pad(i)].
    template <int NF>
    __device__ static __forceinline__ void run(cf (&v)[NF][E_], cf* const (&lds)[NF], const cf* tw, int t) {
        NoStamp st;
        run<NF>(v, lds, tw, t, st, t);
    }
    // stamps (timing builds): 2P+1 = pass P's operands read back and twiddled, 2P+2 = its butterflies done.
    // t0: the lane's first-pass column (v[f][b*R0 + q] = z_f[(t0 + b*LPF) + q*NC/R0]); any permutation of the
    // lanes works there (frame_col_of_lane) because pass 0 only uses it to place its outputs.
    // HALF: the last pass stores only the upper half of each lane's outputs (m >= E/2).  The R2C split pairs
    // Z[k], k = t + i*LPF (i < E/2) — which the lane still holds in registers, reg_of_spectrum(i) — with Z[NC-k],
    // which is always one of the upper-half outputs of lane (LPF - t) mod LPF: the lower half never needs to travel.
    template <int NF, class ST, bool HALF = false>
    __device__ static __forceinline__ void run(cf (&v)[NF][E_], cf* const (&lds)[NF], const cf* tw, int t, ST& st,
                                               int t0) {
        pass<0, NF, ST, HALF>(v, lds, tw, t, st, t0);
        wave_lds_fence();
    }
    // register index (within v[f]) of Z[t + i*LPF] after the last pass: output m = b + NB*k sits in v[b*R + k]
    __device__ static constexpr int reg_of_spectrum(int i) {
        constexpr int R = radix_at(NC_, num_passes(NC_) - 1), NB = E_ / R;
        return (i % NB) * R + (i / NB);
    }
    template <int NF, class ST>
    __device__ static __forceinline__ void run(cf (&v)[NF][E_], cf* const (&lds)[NF], const cf* tw, int t, ST& st) {
        run<NF>(v, lds, tw, t, st, t);
    }

    // ---- one pass in three separately callable pieces (single frame): the streaming kernels interleave the pieces
    // of two frames so that one frame's LDS round trip runs behind the other frame's butterflies.
    // (1) operands of pass P (> 0) back from the exchange area into first-index order v[b*R + q]
    template <int P>
    __device__ static __forceinline__ void pass_readback(cf (&v)[E_], const cf* lds, int t) {
        constexpr int R = radix_at(NC, P), NB = E / R;
        static_assert(P > 0 && (NC / R) % 16 == 0, "read stride must keep the padding affine");
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // NC/R is a multiple of 16 for every pass after the first, so pad(j + c) = pad(j) + pad(c):
            // one address register per butterfly, the rest are DS immediate offsets.
            const cf* src = lds + lds_pad(t + b * LPF);
#pragma unroll
            for (int q = 0; q < R; ++q) v[b * R + q] = src[lds_pad_c(q * (NC / R))];
        }
    }
    // (2) inter-pass twiddles (P > 0) and the radix-R butterflies of pass P
    // REL: `tw` points at pass P's own twiddles (kernels that keep a pass's set somewhere else than the hoisted array)
    template <int P, bool REL = false>
    __device__ static __forceinline__ void pass_twiddle(cf (&v)[E_], const cf* tw) {
        constexpr int R = radix_at(NC, P), OFF = REL ? 0 : twiddles_before(NC, E, P), NB = E / R;
        if constexpr (P > 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const cf* twb = tw + OFF + (pass_shares_twiddles(NC, E, P) ? 0 : b) * (R - 1);
#pragma unroll
                for (int q = 1; q < R; ++q) v[b * R + q] = last_pass_const<P>(v[b * R + q], b * q * (32 / E));
#pragma unroll
                for (int q = 1; q + 1 < R; q += 2) cmul_x2(v[b * R + q], twb[q - 1], v[b * R + q + 1], twb[q]);   // interleaved pairs
                if constexpr (((R - 1) & 1) != 0) v[b * R + R - 1] = cmul(v[b * R + R - 1], twb[R - 2]);
            }
        }
    }
    template <int P>
    __device__ static __forceinline__ void pass_butterflies(cf (&v)[E_]) {
        constexpr int R = radix_at(NC, P), NB = E / R;
        static_assert(NB >= 1, "radix larger than elements per lane");
#pragma unroll
        for (int b = 0; b < NB; ++b) Dft<R>::run(&v[b * R]);
    }
    // (3) outputs of pass P into the exchange area (HALF: the last pass keeps its lower-half outputs in registers)
    template <int P, bool HALF>
    __device__ static __forceinline__ void pass_write(const cf (&v)[E_], cf* lds, int t, int t0) {
        constexpr int KLO = (HALF && pass_is_last(NC_, P)) ? radix_at(NC_, P) / 2 : 0;     // first output stored
        constexpr int R = radix_at(NC, P), S = stride_at(NC, P), NB = E / R;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = (P == 0 ? t0 : t) + b * LPF;
            if constexpr (S == 1) {
                static_assert(R == 16, "first pass is radix 16");
                cf* dst = lds + 17 * j;                          // pad(16 j + k) = 17 j + k, k < 16
#pragma unroll
                for (int k = KLO; k < R; ++k) dst[k] = v[b * R + k];
            } else {
                static_assert(S % 16 == 0, "write stride must keep the padding affine");
                cf* dst = lds + lds_pad((j / S) * (S * R) + (j & (S - 1)));
#pragma unroll
                for (int k = KLO; k < R; ++k) dst[lds_pad_c(k * S)] = v[b * R + k];
            }
        }
    }

    // In-register form of the exchange between passes 1 and 2 of the 16 . 16 . 4 plan (NC = 1024, one frame per wave):
    // the radix-4 butterfly b of lane t = u + 16 a needs outputs a + 4b of the four lanes u + 16 q, i.e. per b a 4 x 4
    // transpose between four registers and the four 16-lane rows of the wave — two v_permlane32_swap and two
    // v_permlane16_swap per dword instead of 16 LDS stores, 16 LDS loads and their round trip.
    __device__ static __forceinline__ void swap_rows32(cf& x, cf& y) {       // x rows {2,3} <-> y rows {0,1}
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
        x = mkc(__uint_as_float(a[0]), __uint_as_float(b[0]));
        y = mkc(__uint_as_float(a[1]), __uint_as_float(b[1]));
    }
    __device__ static __forceinline__ void swap_rows16(cf& x, cf& y) {       // x rows {1,3} <-> y rows {0,2}
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
        x = mkc(__uint_as_float(a[0]), __uint_as_float(b[0]));
        y = mkc(__uint_as_float(a[1]), __uint_as_float(b[1]));
    }
    __device__ static __forceinline__ void exchange_1_2_in_registers(cf (&v)[E_]) {
        static_assert(NC_ == 1024 && E_ == 16 && radix_at(NC_, 1) == 16 && radix_at(NC_, 2) == 4,
                      "wired for the 16 . 16 . 4 plan with 64 lanes per frame");
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            swap_rows32(v[4 * b + 0], v[4 * b + 2]);
            swap_rows32(v[4 * b + 1], v[4 * b + 3]);
            swap_rows16(v[4 * b + 0], v[4 * b + 1]);
            swap_rows16(v[4 * b + 2], v[4 * b + 3]);
        }
    }

    // R2C partners without the exchange area (one frame per wave, LPF == 64).  After the last pass lane t holds
    // Z[t + 64 i] in v[reg_of_spectrum(i)], i < 16.  The partner of pair p, Z[NC - (t + 64 p)] = Z[(64 - t) + 64 (15 - p)], is
    // register reg_of_spectrum(15 - p) of lane 64 - t — the SAME register in every lane — so the eight partners are sixteen
    // ds_bpermute_b32 (the LDS crossbar routes lane to lane; no LDS memory, no bank conflicts, one trip) instead of the
    // half write (8 ds_write_b64) + 9 ds_read_b64 through the area.  Lane 0 pairs within itself: NC - 64 p = 64 (16 - p),
    // its own register reg_of_spectrum(16 - p) (p >= 1), and bin 0 with itself; zmid = Z[NC / 2] is lane 0's register 8.
    __device__ static __forceinline__ void r2c_partners_bpermute(const cf (&v)[E_], cf (&zm)[E_ / 2], cf& zmid, int t) {
        static_assert(LPF == 64 && E_ == 16, "one frame per wave, sixteen elements per lane");
        const int from = ((64 - t) & 63) * 4;               // byte index of the source lane
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const cf src = v[reg_of_spectrum(15 - p)];
            zm[p] = mkc(__int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(src.x))),
                        __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(src.y))));
        }
        if (t == 0) {
            zm[0] = v[reg_of_spectrum(0)];
#pragma unroll
            for (int p = 1; p < NPAIR; ++p) zm[p] = v[reg_of_spectrum(16 - p)];
        }
        zmid = v[reg_of_spectrum(8)];                       // (meaningful in lane 0, the only lane that uses it)
    }

    template <int P, int NF, class ST, bool HALF = false>
    __device__ static __forceinline__ void pass(cf (&v)[NF][E_], cf* const (&lds)[NF], const cf* tw, int t, ST& st,
                                                int t0) {
        if constexpr (P > 0) {
            wave_lds_fence();
#pragma unroll
            for (int f = 0; f < NF; ++f) pass_readback<P>(v[f], lds[f], t);
#pragma unroll
            for (int f = 0; f < NF; ++f) pass_twiddle<P>(v[f], tw);
            st.mark(2 * P + 1);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) pass_butterflies<P>(v[f]);
        st.mark(2 * P + 2);
        wave_lds_fence();   // every lane's reads of this pass precede any lane's writes (same wave, in order)
#pragma unroll
        for (int f = 0; f < NF; ++f) pass_write<P, HALF>(v[f], lds[f], t, t0);
        if constexpr (P + 1 < NPASS) pass<P + 1, NF, ST, HALF>(v, lds, tw, t, st, t0);
    }

    // R2C split of pair index k (0 <= k <= NC/2): returns 2·X[k] in xa and 2·X[NC-k] in xb (the halving of the
    // even/odd split is left to the caller's epilogue factor).  wk = exp(-2*pi*i*k/N).
    //   ev = zk + conj(zm), d = zk - conj(zm), tw = wk·(-i·d);  2X[k] = ev + tw, 2X[NC-k] = conj(ev - tw)
    __device__ static __forceinline__ void r2c_pair(const cf* lds, int k, cf wk, cf& xa, cf& xb) {
        r2c_split_x2(lds[lds_pad(k)], lds[lds_pad((NC - k) & (NC - 1))], wk, xa, xb);
    }
    // partner Z[NC-k] of pair index k = t + i*LPF from LDS; k == 0 pairs with itself (its slot is not stored under HALF)
    __device__ static __forceinline__ cf r2c_partner(const cf* lds, int k, cf zk) {
        const cf zm = lds[lds_pad((NC - k) & (NC - 1))];
        return k == 0 ? zk : zm;
    }
    __device__ static __forceinline__ void r2c_split_x2(cf zk, cf zm, cf wk, cf& xa, cf& xb) {
        const cf ev = cadd_conj(zk, zm), d = csub_conj(zk, zm);
        const cf tw = cmul_rot(wk, d);
        xa = cadd(ev, tw);
        xb = csub_then_conj(ev, tw);
    }
    // (|2X[k]|^2, |2X[NC-k]|^2) of the same split, without forming the two spectra
    __device__ static __forceinline__ cf r2c_power_x2(cf zk, cf zm, cf wk) {
        const cf ev = cadd_conj(zk, zm), d = csub_conj(zk, zm);
        return power_pair(ev, cmul_rot(wk, d));
    }
    __device__ static __forceinline__ cf r2c_power_factored_x2(cf zk, cf zm, cf w0, int i) {
        static_assert(E == 16, "factored R2C twiddles are wired for 2E = 32");
        const cf ev = cadd_conj(zk, zm), d = csub_conj(zk, zm);
        return power_pair(ev, (i == 0) ? cmul_rot(w0, d) : cmul(mul_w32(d, i + 8), w0));
    }
    // Pair index i uses W_N^{t + i*LPF} = W_N^t · W_{2E}^i: one lane-dependent register (w0 = W_N^t) and a
    // COMPILE-TIME constant per pair instead of E/2 hoisted twiddles (E == 16 only: W_32^i; -i = W_32^8).
    __device__ static __forceinline__ void r2c_split_factored_x2(cf zk, cf zm, cf w0, int i, cf& xa, cf& xb) {
        static_assert(E == 16, "factored R2C twiddles are wired for 2E = 32");
        const cf ev = cadd_conj(zk, zm), d = csub_conj(zk, zm);
        const cf tw = (i == 0) ? cmul_rot(w0, d) : cmul(mul_w32(d, i + 8), w0);
        xa = cadd(ev, tw);
        xb = csub_then_conj(ev, tw);
    }
};

// ---------------------------------------------------------------- gradient helpers (backward.hip, stft_n400.hip)
// sources of the inverse transform's gradient spectrum: given, formed from the spectrum and the gradient of |z|^power, or
// with the spectrum itself recomputed from the waveform inside the kernel
enum { SRC_GRAD = 0, SRC_NORM = 1, SRC_WAVE = 2 };

// d/dz of |z|^power (norm then pow, functional.py:126-128): g * power * |z|^(power-2) * z, 0 at z == 0
__device__ __noinline__ inline float norm_pow_factor_general(float s, float power) {      // one copy of powf per kernel
    return power * powf(sqrtf(s), power - 2.0f);
}
template <bool POW2 = false>
__device__ __forceinline__ cf norm_pow_grad(cf v, float gout, float power) {
    const float s = v.x * v.x + v.y * v.y;
    float f;
    if (POW2 || power == 2.0f) f = 2.0f;
    else if (s == 0.0f) f = 0.0f;
    else if (power == 1.0f) f = 1.0f / sqrtf(s);
    else f = norm_pow_factor_general(s, power);
    return cscale(v, f * gout);
}

// Operand of the inverse transform from a pair of the gradient spectrum:  conj(Z[k]) = (conj H[k] + H[NC-k]) - i w_k (conj H[k]
// - H[NC-k]), on the packed-f32 helpers of fft_core.hpp (5 instructions; written out on scalar components it was 14)
__device__ __forceinline__ cf c2r_operand(cf hk, cf hm, cf wk) {
    const cf s = cadd_conj(hm, hk);               // H[NC-k] + conj H[k]
    const cf nd = csub_conj(hm, hk);              // H[NC-k] - conj H[k]
    return csub_rot(s, cmul(nd, wk));             // s + i w (H[NC-k] - conj H[k])
}


// ---------------------------------------------------------------- framing
enum { PAD_CONSTANT = 0, PAD_REFLECT = 1, PAD_REPLICATE = 2, PAD_CIRCULAR = 3 };

struct FrameGeom {
    const float* wave;      // device, rows x row_stride
    long long row_stride;
    long long length;       // L
    const float* window;    // device, win_length
    int win_length;
    int win_offset;         // (N - win_length) / 2
    int hop;
    int center_pad;         // N/2 if center else 0
    int pad_mode;
    int vec2_ok;            // host-verified: cf loads of interior frames are 8-byte aligned
    int vec4_ok;            // ... and 16-byte aligned (hop, pad, row stride multiples of 4 samples)
    long long n_frames;     // T
    long long rows;
    float scale;            // 1 or N^-0.5
};

// zero-padded, centred window value pair for complex element m (samples 2m, 2m+1).  Branch-free:
// clamped unconditional loads + selects, so the 2E loads of a lane issue back to back.
__device__ __forceinline__ cf window_pair(const FrameGeom& g, int m) {
    const int n0 = 2 * m - g.win_offset, n1 = n0 + 1;
    const int last = g.win_length - 1;
    const int c0 = n0 < 0 ? 0 : (n0 > last ? last : n0);
    const int c1 = n1 < 0 ? 0 : (n1 > last ? last : n1);
    const float w0 = g.window[c0], w1 = g.window[c1];
    return mkc(c0 == n0 ? w0 : 0.0f, c1 == n1 ? w1 : 0.0f);
}

// Source index of padded position i (torch.nn.functional.pad semantics), branch-free; *zero is set when
// the sample is a constant-pad zero.  L < 2^31 (host-checked).
__device__ __forceinline__ int padded_index(int i, int L, int mode, bool* zero) {
    const int refl = i < 0 ? -i : (i >= L ? 2 * (L - 1) - i : i);
    const int clmp = i < 0 ? 0 : (i >= L ? L - 1 : i);
    const int circ = i < 0 ? i + L : (i >= L ? i - L : i);
    int j = mode == PAD_REFLECT ? refl : (mode == PAD_CIRCULAR ? circ : clmp);
    j = j < 0 ? 0 : (j >= L ? L - 1 : j);                 // never read out of bounds, whatever the geometry
    *zero = (mode == PAD_CONSTANT) && (i != clmp);
    return j;
}

// Load + window one frame into first-pass register order.  `frame` may be >= T (then zeros).
// Control flow is kept at WHOLE-FRAME granularity on purpose (three straight-line bodies): per-element
// conditions make hipcc split the unrolled loads into one basic block each, which serialises their
// latencies.  Interior frames: 16 cf loads from a wave-uniform base.  Frames touching the padding:
// gathered through padded_index() four elements at a time in a rolled loop via the frame's LDS buffer.
// Sample access of the gather path: float32 waveforms by default; the coded-input kernels (int16 PCM, mu-law codes)
// pass a functor that converts on the way in (melspec_stream.hpp).
struct FetchF32 {
    const float* base;
    __device__ __forceinline__ float operator()(long long row_offset, int j) const { return base[row_offset + j]; }
};

template <class F, bool HOIST_WIN>
__device__ __forceinline__ void load_frame(cf* v, const FrameGeom& g, const cf* win, cf* lds, long long row,
                                           long long frame, int t) {
    load_frame<F, HOIST_WIN, false>(v, g, win, lds, row, frame, t, FetchF32{g.wave});
}

// GATHER_ONLY: never take the vectorised interior path (which reads g.wave as float pairs)
// RAW: deliver the (padded) samples UNwindowed — the caller multiplies by the window exactly as it does for interior frames,
// so that a frame's values do not depend on which path fetched it
template <class F, bool HOIST_WIN, bool GATHER_ONLY, bool RAW = false, class Fetch>
__device__ __forceinline__ void load_frame(cf* v, const FrameGeom& g, const cf* win, cf* lds, long long row,
                                           long long frame, int t, Fetch fetch) {
    constexpr int R0 = radix_at(F::NC, 0);
    constexpr int NB = F::E / R0;
    const float* rp = g.wave + row * g.row_stride;
    const long long row_offset = row * g.row_stride;
    const long long start = frame * (long long)g.hop - g.center_pad;
    if (frame >= g.n_frames) {
#pragma unroll
        for (int e = 0; e < F::E; ++e) v[e] = mkc(0.0f, 0.0f);
    } else if (!GATHER_ONLY && g.vec2_ok && start >= 0 && start + F::N <= g.length) {
        const cf* src = reinterpret_cast<const cf*>(rp + start);
        cf s[F::E];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < R0; ++q) s[b * R0 + q] = src[t + b * F::LPF + q * (F::NC / R0)];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const cf w = HOIST_WIN ? win[b * R0 + q] : window_pair(g, t + b * F::LPF + q * (F::NC / R0));
                v[b * R0 + q] = cmul_elem(s[b * R0 + q], w);
            }
    } else {
        const int L = (int)g.length;
        const int s0 = (int)start;
#pragma unroll 1
        for (int e0 = 0; e0 < F::E; e0 += 4) {
            int j[8];
            bool z[8];
            int mm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u;
                mm[u] = t + (e / R0) * F::LPF + (e % R0) * (F::NC / R0);
                j[2 * u] = padded_index(s0 + 2 * mm[u], L, g.pad_mode, &z[2 * u]);
                j[2 * u + 1] = padded_index(s0 + 2 * mm[u] + 1, L, g.pad_mode, &z[2 * u + 1]);
            }
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = fetch(row_offset, j[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const cf w = RAW ? mkc(1.0f, 1.0f) : window_pair(g, mm[u]);
                lds[lds_pad(mm[u])] = mkc(z[2 * u] ? 0.0f : a[2 * u] * w.x, z[2 * u + 1] ? 0.0f : a[2 * u + 1] * w.y);
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < R0; ++q) v[b * R0 + q] = lds[lds_pad(t + b * F::LPF + q * (F::NC / R0))];
        wave_lds_fence();
    }
}

// ---- split-phase frame load: issue the raw (unwindowed) loads of an interior frame early, apply the window when the
// frame is consumed.  Returns false (nothing loaded) for frames that touch the padding or lie past the end — those go
// through load_frame() at consumption time.  `frame` is per lane group when several frames share a wave (G > 1): the
// request is made for all of the wave's frames or for none (wave ballot), so the caller's fast / gather decision
// stays wave-uniform.  (Fetching with 16-byte requests + v_permlane32_swap measured neutral and was dropped:
// tools/ablation/.)
template <class F>
__device__ __forceinline__ bool prefetch_frame_raw_x(cf* raw, const FrameGeom& g, long long row, long long frame, int t) {
    static_assert(radix_at(F::NC, 0) == F::E, "one first-pass butterfly per lane");
    const long long start = frame * (long long)g.hop - g.center_pad;
    bool ok = g.vec2_ok && frame < g.n_frames && start >= 0 && start + F::N <= g.length;
    if constexpr (F::G > 1) ok = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
    if (ok) {
        const cf* src = reinterpret_cast<const cf*>(g.wave + row * g.row_stride + start);
#pragma unroll
        for (int q = 0; q < F::E; ++q) raw[q] = src[t + q * F::LPF];
    }
    return ok;
}

template <class F>
__device__ __forceinline__ void apply_window(cf* v, const cf* raw, const cf* win) {
#pragma unroll
    for (int e = 0; e < F::E; ++e) v[e] = cmul_elem(raw[e], win[e]);
}

template <class F>
__device__ __forceinline__ void load_window_regs(cf* win, const FrameGeom& g, int t) {
    constexpr int R0 = radix_at(F::NC, 0);
    constexpr int NB = F::E / R0;
    if (g.win_length == F::N) {          // full-length window (the default): plain cf loads, no index math
        const cf* w2 = reinterpret_cast<const cf*>(g.window);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < R0; ++q) win[b * R0 + q] = w2[t + b * F::LPF + q * (F::NC / R0)];
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < R0; ++q) win[b * R0 + q] = window_pair(g, t + b * F::LPF + q * (F::NC / R0));
    }
}

// |X|^power of an (already scaled) spectrum value
__device__ __forceinline__ float cpow_mag(cf x, float power) {
    float s = cnorm2(x);
    if (power == 2.0f) return s;
    float m = sqrtf(s);
    if (power == 1.0f) return m;
    return powf(m, power);
}

// amplitude_to_db: 10*(log10(max(x*x, amin)) - log10(ref))   (reference squares its input)
__device__ __forceinline__ float amp_to_db(float x, float amin, float log10_ref) {
    const float p = x * x;
    const float sq = (p == p) ? fmaxf(p, amin) : p;   // keep NaN: torch.clamp propagates it, fmaxf alone would return amin
    return 10.0f * (log10f(sq) - log10_ref);
}

// the same with the hardware base-2 logarithm (v_log_f32, ~1 ulp of log2): 10 log10(s) = 10 log10(2) log2(s).  The
// hardware flushes denormal inputs, so callers take this form only when amin is a normal number (the clamp then keeps
// the argument normal); absolute error ~1e-6 dB, against ~25 instructions for the correctly rounded log10f.
__device__ __forceinline__ float amp_to_db_fast(float x, float amin, float ten_log10_ref) {
    const float p = x * x;
    const float sq = (p == p) ? fmaxf(p, amin) : p;
    return __builtin_fmaf(3.0102999566398120f, __builtin_amdgcn_logf(sq), -ten_log10_ref);
}

}  // namespace tac
