// melspec_mfma.hpp — the fused fft_length-2048 chain (STFT -> |X|^p -> band-sparse mel filterbank -> dB, one launch) with the
// 1024-point complex transform of every frame on the MATRIX pipe (round 6).  Replaces reference layers.py:307-381 /
// functional.py:99-107, 126-128, 172-184, 291-296 for this size, like melspec_stream3_kernel, whose VALU transform it replaces.
//
// Why.  melspec_stream3_kernel is bound by its instruction streams: 485 VALU + 109 LDS wave-instructions per frame, VALU 74 %
// busy, matrix pipe idle (profiles/r05/pmc_mel.json); its own bounds leave ~6 %.  A 1024-point transform is two chained
// 32 x 32 complex DFT products with a twiddle in between,
//
//     n = 32 n1 + n2,  k = k1 + 32 k2
//     Y[n2][k1]  = sum_n1 z[32 n1 + n2] W32^(n1 k1)              step 1: A = data (row n2 = lane & 31), B = constant
//     Y'[n2][k1] = Y[n2][k1] W1024^(n2 k1)                        VALU, on the accumulator registers
//     Z[k1 + 32 k2] = sum_n2 W32^(k2 n2) Y'[n2][k1]               step 2: A = constant, B = data (column k1 = lane & 31)
//
// and v_mfma_f32_32x32x16_f16 does a 32 x 32 x 16 real product in 32 cycles per SIMD.  float32 accuracy comes from splitting
// every operand into an fp16 (hi, lo) pair — hi = rne16(x), lo = rne16(x - hi): 22+ significant bits — and accumulating the three
// products hi.hi + hi.lo + lo.hi in float32: 4 real products x 3 split terms x 2 K-chunks = 24 MFMAs per step, 48 per frame =
// 1 536 matrix-pipe cycles against ~1 900 VALU cycles of the radix-16 form, and they overlap with the VALU work of the SIMD's
// other waves (window, split, twiddle, R2C, contraction).  Against a float64 transform the result is within 2.5e-7 of the
// spectrum's maximum (numpy's own float32 transform: 1.2e-7) — tools/emulate_mfma_fft.py emulates the data flow lane-accurately.
//
// No exchange through the LDS inside the transform: the D layout of the 32 x 32 MFMAs (column = lane & 31, register r of lane
// half h = row 8 (r >> 2) + 4 h + (r & 3)) is made the K order of BOTH steps' constant operand (mf_perm), so step 1's
// accumulators, twiddled and split in place, ARE step 2's B operand.  Step 2 leaves Z[k1 + 32 k2] with the lower half of the
// spectrum (k < 512) in registers 0..7 and the upper half in 8..15 of every lane; the R2C partner of a lower bin is an upper bin
// of one other lane, so the upper half crosses once through a 4 KB area (planar: real and imaginary parts apart, two bins per
// packed instruction), which the |X|^p row then overwrites — 4.2 KB of LDS per wave instead of 8.7.
//
// fp16 range: a frame is scaled by a power of two so that its largest windowed sample lies in [2^7, 2^8) (one wave-wide max per
// frame; |Y'| <= 32 sqrt(2) 2^8 < 65 504), and the mel row is scaled back (exactly) before the dB epilogue.
#pragma once
#include "melspec_stream3.hpp"

namespace tac {

typedef _Float16 mf_h8 __attribute__((ext_vector_type(8)));
typedef float mf_acc __attribute__((ext_vector_type(16)));
typedef unsigned mf_u4 __attribute__((ext_vector_type(4)));

constexpr int MF_CB = 4;         // contraction steps per LDS round trip in the unrolled (FAST1) form (7: +13 %, spills)

// K slot (MFMA c of a step's two, lane half h, element j of the lane's eight) -> summation index: the row of accumulator
// register 8 c + j in lane half h
__host__ __device__ constexpr int mf_perm(int c, int h, int j) { return 16 * c + 8 * (j >> 2) + 4 * h + (j & 3); }

// ---- float32 -> fp16 (hi, lo) pairs.  v_fma_mix{lo,hi}_f16: fma in float32, result rounded to fp16 into one half of the
// destination (the other half is kept); sources are float32 (op_sel_hi 0) or one half of a packed fp16 pair (op_sel_hi 1)
__device__ __forceinline__ unsigned mf_pack_hi(float a, float b, float sc) {
    unsigned h;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(a), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h) : "v"(b), "s"(sc));
    return h;
}
__device__ __forceinline__ unsigned mf_pack_lo(float a, float b, float sc, unsigned h) {      // rne16(a sc - hi), exact difference
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(l) : "v"(a), "s"(sc), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "s"(sc), "v"(h));
    return l;
}
// unscaled forms (step 2: |Y'| < 2^14 by construction)
__device__ __forceinline__ unsigned mf_pack_hi1(float a, float b) {
    unsigned h;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    return h;
}
__device__ __forceinline__ unsigned mf_pack_lo1(float a, float b, unsigned h) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(l) : "v"(a), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(h));
    return l;
}
__device__ __forceinline__ void mf_mma(mf_acc& d, mf_u4 a, mf_u4 b) {
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_h8, a), __builtin_bit_cast(mf_h8, b), d, 0, 0, 0);
}

// The constant operand of both steps: W32 (symmetric) as fp16 (hi, lo) pairs in operand layout — lane (i = lane & 31, h = lane >> 5)
// holds W32[i][mf_perm(c, h, j)], j < 8, for the two MFMAs c of a step; as B of step 1 that is column i, as A of step 2 row i.
// n*: the negated imaginary part (the real part of a complex product subtracts).  48 registers for the kernel's lifetime.
struct MfConsts {
    mf_u4 rh[2], rl[2], ih[2], il[2], nih[2], nil[2];
    __device__ __forceinline__ void load(const cf* __restrict__ w_nc, int lane) {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const cf w0 = w_nc[32 * ((mf_perm(c, h, 2 * jj) * i) & 31)];           // W32^(n i) = W1024^(32 (n i mod 32))
                const cf w1 = w_nc[32 * ((mf_perm(c, h, 2 * jj + 1) * i) & 31)];
                rh[c][jj] = mf_pack_hi1(w0.x, w1.x);
                rl[c][jj] = mf_pack_lo1(w0.x, w1.x, rh[c][jj]);
                ih[c][jj] = mf_pack_hi1(w0.y, w1.y);
                il[c][jj] = mf_pack_lo1(w0.y, w1.y, ih[c][jj]);
                nih[c][jj] = ih[c][jj] ^ 0x80008000u;
                nil[c][jj] = il[c][jj] ^ 0x80008000u;
            }
    }
};

// one step = 24 MFMAs: (dr, di) += data x constant with the data as A (DATA_IS_A, step 1) or as B (step 2); every accumulator's
// twelve products back to back (the same-accumulator forwarding path: -4 ... -8 % against alternating the two, profiles/r06/ab/),
// large terms first
template <bool DATA_IS_A>
__device__ __forceinline__ void mf_step(mf_acc& dr, mf_acc& di, const MfConsts& K, const mf_u4 (&xrh)[2], const mf_u4 (&xrl)[2],
                                        const mf_u4 (&xih)[2], const mf_u4 (&xil)[2]) {
    auto mm = [](mf_acc& d, mf_u4 data, mf_u4 con) {
        if constexpr (DATA_IS_A) mf_mma(d, data, con);
        else mf_mma(d, con, data);
    };
#pragma unroll
    for (int c = 0; c < 2; ++c) { mm(dr, xrh[c], K.rh[c]); mm(dr, xih[c], K.nih[c]); }
#pragma unroll
    for (int c = 0; c < 2; ++c) { mm(dr, xrh[c], K.rl[c]); mm(dr, xrl[c], K.rh[c]); mm(dr, xih[c], K.nil[c]); mm(dr, xil[c], K.nih[c]); }
#pragma unroll
    for (int c = 0; c < 2; ++c) { mm(di, xrh[c], K.ih[c]); mm(di, xih[c], K.rh[c]); }
#pragma unroll
    for (int c = 0; c < 2; ++c) { mm(di, xrh[c], K.il[c]); mm(di, xrl[c], K.ih[c]); mm(di, xih[c], K.rl[c]); mm(di, xil[c], K.rh[c]); }
}

// complex element of the frame held in sample register q of lane `lane`: m = 32 n1 + n2, n2 = lane & 31, n1 = mf_perm(q >> 3, lane >> 5, q & 7)
__host__ __device__ constexpr int mf_elem_const(int q) { return 32 * (16 * (q >> 3) + 8 * ((q & 7) >> 2) + ((q & 7) & 3)); }
__device__ __forceinline__ int mf_elem_lane(int lane) { return (lane & 31) + 128 * (lane >> 5); }

// bytes of one wave's LDS area: the partner planes (2 x 512 floats), then, in place, the |X|^p row + its slack taps
template <int NC, int E>
__host__ __device__ constexpr int mf_area_bytes() {
    return (StreamCfg<NC, E>::PROW * 4 + 127) & ~127;
}
// areas + bank weights + twiddle table + window table + frame counter
template <int NC, int E>
__host__ __device__ inline size_t mfma_lds_bytes(int wtot, int waves) {
    return (size_t)waves * mf_area_bytes<NC, E>() + (((size_t)wtot * 4 + 15) & ~(size_t)15) + 8192 + 8192 + 16;
}

// FAST1: as melspec_stream3_kernel (0 = any bank the lane layout takes, else the steps of slot 1 of the (4, FAST1) two-slot layout)
template <bool POW2, int FAST1, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (WAVES + 3) / 4)
melspec_mfma_kernel(FrameGeom g, Tables tb, StreamArgs m) {
    constexpr int NC = 1024, E = 16;
    using C = StreamCfg<NC, E>;
    constexpr int NBINS = C::NBINS;
    constexpr int XA_BYTES = mf_area_bytes<NC, E>();
    constexpr int THREADS = WAVES * 64;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const prow = reinterpret_cast<float*>(smem_raw + (size_t)w * XA_BYTES);   // partner planes, then the |X|^p row
    const long long chunk = m.chunk;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < m.total ? begin + chunk : m.total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    const unsigned T = (unsigned)g.n_frames;
    const int k1 = lane & 31, hh = lane >> 5;
    const int el = mf_elem_lane(lane);

    const int deal_shift = nloc >> 1;                      // edge frames are dealt in the middle of the run (melspec_stream3.hpp)
    auto place = [&](int i) { const int j = i + deal_shift; return j < nloc ? j : j - nloc; };
    cf v[E];
    int mode = 0, row = 0;
    long long fr = 0;
    auto request = [&](int i) {
        i = i < nloc ? i : nloc - 1;
        const unsigned gf = (unsigned)(begin + place(i));
        const unsigned r = gf / T;
        row = (int)r;
        fr = (long long)(gf - r * T);
        const long long start = fr * (long long)g.hop - g.center_pad;
        const bool ok = g.vec2_ok && start >= 0 && start + 2 * NC <= g.length;
        mode = ok ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + 2 * NC <= g.length ? cs : g.length - 2 * NC;
        const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)row * g.row_stride + cs) + el;
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = src[mf_elem_const(q)];
    };

    // ---- tables.  Every global load of the set-up is issued before the first LDS store.
    float* const wlds = reinterpret_cast<float*>(smem_raw + (size_t)WAVES * XA_BYTES);
    f4* const twl = reinterpret_cast<f4*>(wlds + ((m.wtot + 3) & ~3));     // [u < 8][lane]: (re r, re r + 1, im r, im r + 1) of W1024^(n2 k1), r = 2 u
    cf* const winl = reinterpret_cast<cf*>(twl + 512);                    // [q >> 1][lane][q & 1]: window pair of sample register q
    unsigned* const next_frame = reinterpret_cast<unsigned*>(winl + 1024);
    const float half = 0.5f * g.scale;                     // the R2C split returns 2X: folded into the window
    constexpr int WCH = 4;
    const int n4 = (m.wtot + 3) >> 2;
    pf4 wreg[WCH];
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const int c = tid + u * THREADS;
        wreg[u] = reinterpret_cast<const pf4*>(m.wl)[c < n4 ? c : n4 - 1];
    }
    constexpr int NWIN = (1024 + THREADS - 1) / THREADS;
    cf winv[NWIN];
#pragma unroll
    for (int u = 0; u < NWIN; ++u) {
        const int idx = tid + u * THREADS, q = (idx >> 6) & 15, ln = idx & 63;
        const int mel = mf_elem_lane(ln) + 32 * mf_perm(q >> 3, 0, q & 7);       // (the half's 4 h rides in mf_elem_lane's 128 h)
        winv[u] = g.win_length == 2 * NC ? reinterpret_cast<const cf*>(g.window)[mel] : window_pair(g, mel);
    }
    cf twa = mkc(0.f, 0.f), twb = mkc(0.f, 0.f);
    if (tid < 512) {
        const int u = tid >> 6, ln = tid & 63, r = 2 * u;
        const int n2 = 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3);
        twa = tb.w_nc[n2 * (ln & 31)];
        twb = tb.w_nc[(n2 + 1) * (ln & 31)];
    }
    MfConsts K;
    K.load(tb.w_nc, lane);
    // the lane's eight R2C twiddles w_N^k, k = k1 + 32 k2(r), r < 8, planar in register pairs (r, r + 1)
    cf pc[4], ps[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 2 * u;
        const int ka = k1 + 32 * (8 * (r >> 2) + 4 * hh + (r & 3));
        const cf a = tb.w_n[ka], b = tb.w_n[ka + 32];
        pc[u] = mkc(a.x, b.x);
        ps[u] = mkc(a.y, b.y);
    }
    int lo_s[ST_MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < ST_MAX_SLOTS; ++s) lo_s[s] = s < m.nslot ? m.lo[s * 64 + lane] : 0;
    __builtin_amdgcn_sched_barrier(0);
    if (nloc > 0) request(w);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const int c = tid + u * THREADS;
        if (c < n4) reinterpret_cast<pf4*>(wlds)[c] = wreg[u];
    }
    for (int c = tid + WCH * THREADS; c < n4; c += THREADS) reinterpret_cast<pf4*>(wlds)[c] = reinterpret_cast<const pf4*>(m.wl)[c];
    if (tid == 0) *next_frame = WAVES;
#pragma unroll
    for (int u = 0; u < NWIN; ++u) {
        const int idx = tid + u * THREADS, q = (idx >> 6) & 15, ln = idx & 63;
        if (idx < 1024) winl[((q >> 1) * 64 + ln) * 2 + (q & 1)] = cscale(winv[u], half);
    }
    if (tid < 512) {
        f4 x;
        x.x = twa.x; x.y = twb.x; x.z = twa.y; x.w = twb.y;
        twl[tid] = x;
    }
    __syncthreads();
    if (nloc <= 0) return;

    const bool fast_db = m.amin >= 1.1754944e-38f;
    const float ten_log10_ref = 10.0f * m.log10_ref;
    auto fma4 = [](f4 wv, f4 pv, cf& a0, cf& a1) {
        a0 = __builtin_elementwise_fma(mkc(wv.x, wv.y), mkc(pv.x, pv.y), a0);
        a1 = __builtin_elementwise_fma(mkc(wv.z, wv.w), mkc(pv.z, pv.w), a1);
    };
    // float offsets in the wave's area: upper-half bin k - 512 = lo_idx + 256 a + 32 b of register 8 + 4 a + b; the partner of
    // lower register 4 a + b sits at hi_idx - 256 a - 32 b (re plane; im plane 512 floats above)
    const int lo_idx = k1 + 128 * hh;
    float* const pw0 = prow + lo_idx;                      // upper-half writes / lower-half row bins: + 256 a + 32 b
    const float* const pr0 = prow + (416 - lo_idx);        // partner reads: - 256 a, + 96 - 32 b
    float* const ph0 = prow + (928 - lo_idx);              // upper-half row bins 1024 - k: - 256 a, + 96 - 32 b
    int i = w;
    unsigned long long probe_c = 0, probe_w = 0;
    if (m.probe && w == 0) {
        probe_c = __builtin_readcyclecounter();
        probe_w = wall_clock64();
    }
    while (i < nloc) {
        unsigned ask = 0;
        if (lane == 0) ask = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- s0: samples (frames touching the padding gather theirs), window, frame scale, fp16 split
        if (mode != 1) {
            const int L = (int)g.length;
            const int s0 = (int)(fr * (long long)g.hop - g.center_pad);
            const long long row_offset = (long long)row * g.row_stride;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int mm = el + mf_elem_const(q);
                bool z0, z1;
                const int j0 = padded_index(s0 + 2 * mm, L, g.pad_mode, &z0);
                const int j1 = padded_index(s0 + 2 * mm + 1, L, g.pad_mode, &z1);
                const float a0 = g.wave[row_offset + j0], a1 = g.wave[row_offset + j1];
                v[q] = mkc(z0 ? 0.0f : a0, z1 ? 0.0f : a1);
            }
        }
        float mx = 0.0f;
        {
            const f4* wl = reinterpret_cast<const f4*>(winl) + lane;
#pragma unroll
            for (int u = 0; u < E / 2; ++u) {
                const f4 x = wl[u * 64];
                v[2 * u] = v[2 * u] * mkc(x.x, x.y);
                v[2 * u + 1] = v[2 * u + 1] * mkc(x.z, x.w);
            }
#pragma unroll
            for (int q = 0; q < E; ++q) mx = fmaxf(mx, fmaxf(__builtin_fabsf(v[q].x), __builtin_fabsf(v[q].y)));
        }
        float sc, inv;
        {
            // non-negative floats order like integers: wave-wide max over the sixteen lanes of a row by DPP, over the rows by readlane
            int mi = __float_as_int(mx);
            mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0xB1, 0xF, 0xF, true));      // quad_perm [1, 0, 3, 2]
            mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0x4E, 0xF, 0xF, true));      // quad_perm [2, 3, 0, 1]
            mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0x141, 0xF, 0xF, true));     // row_half_mirror
            mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0x140, 0xF, 0xF, true));     // row_mirror
            const int r0 = __builtin_amdgcn_readlane(mi, 0), r1 = __builtin_amdgcn_readlane(mi, 16);
            const int r2 = __builtin_amdgcn_readlane(mi, 32), r3 = __builtin_amdgcn_readlane(mi, 48);
            int ex = max(max(r0, r1), max(r2, r3)) >> 23;       // biased exponent of the frame's largest windowed sample
            ex = ex < 8 ? 8 : ex;
            sc = __int_as_float((261 - ex) << 23);              // 2^(134 - ex): the maximum lands in [2^7, 2^8)
            inv = __int_as_float((ex - 7) << 23);               // 1 / sc
        }
        mf_u4 xrh[2], xrl[2], xih[2], xil[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const cf a = v[8 * c + 2 * jj], b = v[8 * c + 2 * jj + 1];
                xrh[c][jj] = mf_pack_hi(a.x, b.x, sc);
                xih[c][jj] = mf_pack_hi(a.y, b.y, sc);
                xrl[c][jj] = mf_pack_lo(a.x, b.x, sc, xrh[c][jj]);
                xil[c][jj] = mf_pack_lo(a.y, b.y, sc, xih[c][jj]);
            }
        // ---- step 1
        mf_acc dr = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        mf_acc di = dr;
        mf_step<true>(dr, di, K, xrh, xrl, xih, xil);
        // ---- twiddle W1024^(n2 k1) on the accumulators, split: step 2's B operand
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f4 tw = twl[u * 64 + lane];
            const cf twr = mkc(tw.x, tw.y), twi = mkc(tw.z, tw.w);
            const cf ar = mkc(dr[2 * u], dr[2 * u + 1]), ai = mkc(di[2 * u], di[2 * u + 1]);
            const cf yr = __builtin_elementwise_fma(-ai, twi, ar * twr);
            const cf yi = __builtin_elementwise_fma(ai, twr, ar * twi);
            const int c = u >> 2, jj = u & 3;
            xrh[c][jj] = mf_pack_hi1(yr.x, yr.y);
            xih[c][jj] = mf_pack_hi1(yi.x, yi.y);
            xrl[c][jj] = mf_pack_lo1(yr.x, yr.y, xrh[c][jj]);
            xil[c][jj] = mf_pack_lo1(yi.x, yi.y, xih[c][jj]);
        }
        // ---- step 2: lane (k1, h), register r: Z[k1 + 32 (8 (r >> 2) + 4 h + (r & 3))]
        mf_acc zr = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        mf_acc zi = zr;
        mf_step<false>(zr, zi, K, xrh, xrl, xih, xil);
        // ---- s3: the upper half of the spectrum crosses to its R2C partners (planar), R2C split -> |X|^p row in place
        wave_lds_fence();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                pw0[256 * a + 32 * b] = zr[8 + 4 * a + b];
                pw0[512 + 256 * a + 32 * b] = zi[8 + 4 * a + b];
            }
        wave_lds_fence();
        float mr[8], mi8[8];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                mr[4 * a + b] = pr0[96 - 32 * b - 256 * a];
                mi8[4 * a + b] = pr0[512 + 96 - 32 * b - 256 * a];
            }
        if (lane == 0) {                                    // bin 0 pairs with itself
            mr[0] = zr[0];
            mi8[0] = zi[0];
        }
        float plo[8], phi[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = 2 * u;
            const cf kr = mkc(zr[r], zr[r + 1]), ki = mkc(zi[r], zi[r + 1]);
            const cf pr = mkc(mr[r], mr[r + 1]), pi = mkc(mi8[r], mi8[r + 1]);
            // ev = zk + conj(zm), d = zk - conj(zm), tw = w (-i d);  2X[k] = ev + tw, 2X[NC - k] = conj(ev - tw)
            const cf evr = kr + pr, evi = ki - pi, ddr = kr - pr, ddi = ki + pi;
            const cf twr = __builtin_elementwise_fma(pc[u], ddi, ps[u] * ddr);
            const cf twi = __builtin_elementwise_fma(ps[u], ddi, -(pc[u] * ddr));
            const cf ar = evr + twr, ai = evi + twi, br = evr - twr, bi = evi - twi;
            const cf pa = __builtin_elementwise_fma(ai, ai, ar * ar), pb = __builtin_elementwise_fma(bi, bi, br * br);
            plo[r] = POW2 ? pa.x : __builtin_amdgcn_sqrtf(pa.x);
            plo[r + 1] = POW2 ? pa.y : __builtin_amdgcn_sqrtf(pa.y);
            phi[r] = POW2 ? pb.x : __builtin_amdgcn_sqrtf(pb.x);
            phi[r + 1] = POW2 ? pb.y : __builtin_amdgcn_sqrtf(pb.y);
        }
        const float pmid = 4.0f * (zr[8] * zr[8] + zi[8] * zi[8]);        // lane 0: Z[512] is its own register 8
        wave_lds_fence();                                                   // all partner reads are in registers
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                pw0[256 * a + 32 * b] = plo[4 * a + b];
                ph0[96 - 32 * b - 256 * a] = phi[4 * a + b];
            }
        if (lane == 0) prow[NC / 2] = POW2 ? pmid : __builtin_amdgcn_sqrtf(pmid);
        if (lane < C::PROW - NBINS) prow[NBINS + lane] = 0.0f;              // slack taps carry zero weights: keep them finite
        wave_lds_fence();
        // ---- the next frame's samples go out now (the sample registers are dead), they land during the contraction
        const int cur = i;
        i = (int)__builtin_amdgcn_readfirstlane(ask);
        request(i);
        // ---- s4: filterbank contraction, the frame's scale back, dB, row store
        const float inv2 = POW2 ? inv : 1.0f;
        if constexpr (FAST1 > 0) {
            const int ci = cur < nloc ? cur : nloc - 1;
            const f4* wp = reinterpret_cast<const f4*>(wlds) + lane;
            const f4* p0 = reinterpret_cast<const f4*>(prow + lo_s[0]);
            const f4* p1 = reinterpret_cast<const f4*>(prow + lo_s[1]);
            cf a0 = mkc(0.f, 0.f), a1 = mkc(0.f, 0.f), b0 = mkc(0.f, 0.f), b1 = mkc(0.f, 0.f);
            // batches of MF_CB steps (one LDS round trip each): the constant operands and the requested samples leave the
            // contraction ~50 registers, and the SIMD's other waves cover the round trips
            constexpr int CB = MF_CB;
            {
                f4 wv[ST_FAST_STEPS0], qv[ST_FAST_STEPS0];
#pragma unroll
                for (int u = 0; u < ST_FAST_STEPS0; ++u) {
                    wv[u] = wp[u * 64];
                    qv[u] = p0[u];
                }
#pragma unroll
                for (int u = 0; u < ST_FAST_STEPS0; ++u) fma4(wv[u], qv[u], a0, a1);
            }
#pragma unroll
            for (int j0 = 0; j0 < FAST1; j0 += CB) {
                f4 wv[CB], qv[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u)
                    if (j0 + u < FAST1) {
                        wv[u] = wp[(ST_FAST_STEPS0 + j0 + u) * 64];
                        qv[u] = p1[j0 + u];
                    }
#pragma unroll
                for (int u = 0; u < CB; ++u)
                    if (j0 + u < FAST1) fma4(wv[u], qv[u], b0, b1);
            }
            float v0 = ((a0.x + a0.y) + (a1.x + a1.y)) * inv * inv2, v1 = ((b0.x + b0.y) + (b1.x + b1.y)) * inv * inv2;
            if (m.db) {
                v0 = fast_db ? amp_to_db_fast(v0, m.amin, ten_log10_ref) : amp_to_db(v0, m.amin, m.log10_ref);
                v1 = fast_db ? amp_to_db_fast(v1, m.amin, ten_log10_ref) : amp_to_db(v1, m.amin, m.log10_ref);
            }
            float* orow = m.out + (begin + place(ci)) * (long long)m.n_mels + lane;
            orow[0] = v0;
            orow[64] = v1;
        } else {
            const int ci = cur < nloc ? cur : nloc - 1;
            const f4* wp = reinterpret_cast<const f4*>(wlds) + lane;
            float* orow = m.out + (begin + place(ci)) * (long long)m.n_mels + (m.rev ? m.n_mels - 1 - lane : lane);
            const int ostep = m.rev ? -64 : 64;                           // cell 64 s + lane -> its band (StreamArgs::rev)
#pragma unroll
            for (int s = 0; s < ST_MAX_SLOTS; ++s) {
                if (s < m.nslot) {
                    const f4* pp = reinterpret_cast<const f4*>(prow + lo_s[s]);
                    const int n = m.steps[s];
                    cf acc0 = mkc(0.f, 0.f), acc1 = mkc(0.f, 0.f);
                    int j = 0;
#pragma unroll 1
                    for (; j + 4 <= n; j += 4, wp += 256) {
                        f4 wv[4], pv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            wv[u] = wp[u * 64];
                            pv[u] = pp[j + u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) fma4(wv[u], pv[u], acc0, acc1);
                    }
#pragma unroll 1
                    for (; j < n; ++j, wp += 64) fma4(wp[0], pp[j], acc0, acc1);
                    float val = ((acc0.x + acc0.y) + (acc1.x + acc1.y)) * inv * inv2;
                    if (m.db) val = fast_db ? amp_to_db_fast(val, m.amin, ten_log10_ref) : amp_to_db(val, m.amin, m.log10_ref);
                    if (s * 64 + lane < m.n_mels) orow[s * ostep] = val;
                }
            }
        }
        wave_lds_fence();                                                   // the row is consumed: the area takes the next frame
    }
    if (m.probe && w == 0) {
        const unsigned long long dc = __builtin_readcyclecounter() - probe_c, dw = wall_clock64() - probe_w;
        if (lane == 0) {
            m.probe[2 * blockIdx.x] = dc;
            m.probe[2 * blockIdx.x + 1] = dw;
        }
    }
}

}  // namespace tac
