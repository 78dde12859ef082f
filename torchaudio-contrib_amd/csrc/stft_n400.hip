// stft_n400.hip — one-sided STFT / |X|^p (+dB) rows for fft_length 400 (25 ms at 16 kHz, the usual speech front end),
// which is not a power of two: 400 real samples = 200 complex z[n], 200 = 8 · 25.
//
// A frame is held by EIGHT lanes (eight frames per wave); with n = e + 8m and k = k2 + 25·k1
//     Z[k2 + 25 k1] = sum_e W_8^{e k1} · ( W_200^{e k2} · sum_m z[e + 8m] W_25^{m k2} ).
//   (1) lane e transforms its 25 samples z[e + 8m] entirely in registers: 5 × 5 Cooley-Tukey, ten radix-5 butterflies,
//       compile-time W_25 twiddles (no exchange);
//   (2) multiplies by its W_200^{e k2} row (a 208-byte table row in LDS, conflict-free 16-byte reads);
//   (3) the 8-point transform over e runs ACROSS the eight lanes: three radix-2 decimation-in-frequency stages whose
//       partners come through DPP (row_half_mirror, quad_perm) — lane l holds e = l (l < 4) or 11 - l, which turns the
//       pairings e ^ 4, e ^ 2, e ^ 1 into the lane pairings l ^ 7, l ^ 2, l ^ 1 that DPP can do; lane l ends with
//       k1 = bitrev3(e).  No LDS, no barrier.
//   (4) R2C split for the lane's own 25 bins: the partner Z[200 - k] sits in lane l ^ 4 (two DPP moves), register
//       25 - k2; the k2 = 0 column pairs by ds_bpermute.  Bin 200 comes from the lane that holds bin 0.
// The rest is the stft_small.hip recipe: a unit is eight consecutive frames of one row, so its rows are adjacent in
// memory, are staged in LDS in output order and leave as one run of nontemporal 16-byte stores; the next unit's
// samples are requested before the current unit's rows are stored; units whose frames touch the padding (or the end
// of the row) gather sample by sample; one 8-wave workgroup per CU draws units from a workgroup counter.
// Reference: torchaudio_contrib/functional.py:48-113 (stft), :116-128 (complex_norm), :277-296 (amplitude_to_db).
#include "host_common.hpp"
#include "mel_lanes.hpp"
#include "ola_plan.hpp"

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace tac {

constexpr int Q4_WAVES = 8;     // waves per workgroup
constexpr int Q4_G = 8;         // frames per wave
constexpr int Q4_M = 25;        // samples / bins per lane
constexpr int Q4_ROW = 26;      // table row pitch in cf (208 B: eight rows hit eight different 16-byte bank groups)
constexpr int Q4_BINS = 201;
// floats of LDS per wave: the unit's eight complex rows (+ the 16-byte phase), which is also enough for the 64 x 25
// complex values the gather path parks there
constexpr int Q4_STAGE = ((Q4_G * 2 * Q4_BINS + 4 + 3) / 4) * 4;
static_assert(Q4_STAGE >= 64 * Q4_M * 2, "staging area holds a gathered unit");
typedef float q4_f4 __attribute__((ext_vector_type(4)));

struct Q4Tables {
    const cf* w200;             // [8][Q4_ROW]: W_200^{e(l) k2}
    const cf* w400;             // [8][Q4_ROW]: W_400^{25 k1(l) + k2}
    const cf* w400b;            // [8][Q4_ROW]: W_400^{e(l) + 8 m} — the C2R twiddles in the transform's INPUT layout (backward)
};

// The fused Melspectrogram form (MEL): the unit's |X|^p rows stay in LDS (204-float pitch) and are contracted with a
// band-sparse filterbank there by the frame's eight lanes (mel_lanes.hpp); the mel rows are staged behind them.
constexpr int Q4_MEL_PITCH = 204;
#ifndef TAC_Q4_FLY
#define TAC_Q4_FLY 8
#endif
constexpr int Q4_FLY = TAC_Q4_FLY;       // contraction steps in flight (64 registers: the twelve-wave kernel has the next unit's 50 in flight too)
constexpr int Q4_MEL_OFF = Q4_G * Q4_MEL_PITCH + 8;
static_assert(Q4_MEL_OFF + 4 + Q4_G * LM_MAX_MELS <= Q4_STAGE, "mel rows fit the staging area");

__host__ __device__ constexpr int q4_e_of(int l) { return l < 4 ? l : 11 - l; }
__host__ __device__ constexpr int q4_bitrev3(int e) { return ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1); }
__host__ __device__ constexpr int q4_k1_of(int l) { return q4_bitrev3(q4_e_of(l)); }

template <int CTRL>
__device__ __forceinline__ cf q4_dpp(cf a) {
    return mkc(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.x), CTRL, 0xf, 0xf, false)),
               __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.y), CTRL, 0xf, 0xf, false)));
}
constexpr int Q4_HALF_MIRROR = 0x141;                      // lane l <- lane 7 - l of its group of eight
constexpr int Q4_QUAD_XOR1 = 0xB1, Q4_QUAD_XOR2 = 0x4E, Q4_QUAD_XOR3 = 0x1B;   // quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0]

// forward 5-point DFT in place
__device__ __forceinline__ void q4_dft5(cf& x0, cf& x1, cf& x2, cf& x3, cf& x4) {
    constexpr float C1 = 0.30901699437494745f, C2 = -0.80901699437494745f;
    constexpr float S1 = 0.95105651629515353f, S2 = 0.58778525229247314f;
    const cf t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    const cf a1 = __builtin_elementwise_fma(t2, mkc(C2, C2), __builtin_elementwise_fma(t1, mkc(C1, C1), x0));
    const cf a2 = __builtin_elementwise_fma(t2, mkc(C1, C1), __builtin_elementwise_fma(t1, mkc(C2, C2), x0));
    const cf b1 = __builtin_elementwise_fma(t4, mkc(S2, S2), t3 * mkc(S1, S1));
    const cf b2 = __builtin_elementwise_fma(t4, mkc(-S1, -S1), t3 * mkc(S2, S2));
    x0 = cadd(x0, cadd(t1, t2));
    x1 = cadd_rot(a1, b1);                                 // a - i b
    x4 = csub_rot(a1, b1);                                 // a + i b
    x2 = cadd_rot(a2, b2);
    x3 = csub_rot(a2, b2);
}

// W_25^j = (cos, -sin)(2 pi j / 25), j = 0 .. 16
__device__ constexpr float Q4_COS25[17] = {1.000000000e+00f, 9.685831611e-01f, 8.763066800e-01f, 7.289686274e-01f,
                                           5.358267950e-01f, 3.090169944e-01f, 6.279051953e-02f, -1.873813146e-01f,
                                           -4.257792916e-01f, -6.374239897e-01f, -8.090169944e-01f, -9.297764859e-01f,
                                           -9.921147013e-01f, -9.921147013e-01f, -9.297764859e-01f, -8.090169944e-01f,
                                           -6.374239897e-01f};
__device__ constexpr float Q4_SIN25[17] = {0.000000000e+00f, 2.486898872e-01f, 4.817536741e-01f, 6.845471059e-01f,
                                           8.443279255e-01f, 9.510565163e-01f, 9.980267284e-01f, 9.822872507e-01f,
                                           9.048270525e-01f, 7.705132428e-01f, 5.877852523e-01f, 3.681245527e-01f,
                                           1.253332336e-01f, -1.253332336e-01f, -3.681245527e-01f, -5.877852523e-01f,
                                           -7.705132428e-01f};

// 25-point DFT in registers: v[m] -> v[k2], natural order in and out
__device__ __forceinline__ void q4_dft25(cf (&v)[Q4_M]) {
#pragma unroll
    for (int m2 = 0; m2 < 5; ++m2) {                        // over m1 (m = 5 m1 + m2): A[m2][q1] lands in v[5 q1 + m2]
        q4_dft5(v[m2], v[5 + m2], v[10 + m2], v[15 + m2], v[20 + m2]);
#pragma unroll
        for (int q1 = 1; q1 < 5; ++q1)
            if (m2 > 0) v[5 * q1 + m2] = cmulc(v[5 * q1 + m2], Q4_COS25[m2 * q1], -Q4_SIN25[m2 * q1]);
    }
    cf y[Q4_M];
#pragma unroll
    for (int q1 = 0; q1 < 5; ++q1) {                        // over m2: Y[q1 + 5 q2]
        cf b0 = v[5 * q1], b1 = v[5 * q1 + 1], b2 = v[5 * q1 + 2], b3 = v[5 * q1 + 3], b4 = v[5 * q1 + 4];
        q4_dft5(b0, b1, b2, b3, b4);
        y[q1] = b0; y[q1 + 5] = b1; y[q1 + 10] = b2; y[q1 + 15] = b3; y[q1 + 20] = b4;
    }
#pragma unroll
    for (int k = 0; k < Q4_M; ++k) v[k] = y[k];
}

// one 208-byte table row -> 25 register values (twelve 16-byte reads and one 8-byte read)
__device__ __forceinline__ void q4_read_row(const cf* row, cf (&r)[Q4_M]) {
    const q4_f4* p = reinterpret_cast<const q4_f4*>(row);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const q4_f4 x = p[i];
        r[2 * i] = mkc(x.x, x.y);
        r[2 * i + 1] = mkc(x.z, x.w);
    }
    r[24] = row[24];
}

template <int MODE, bool MEL, int S>
__global__ void __launch_bounds__(Q4_WAVES * 64, 2)
stft_n400_kernel(FrameGeom g, Q4Tables tb, StftEpilogue ep, LaneMel mel) {
    constexpr int LENF = (MODE == 0 ? 2 : 1) * Q4_BINS;
    constexpr int STAGE = Q4_STAGE;                                        // floats per wave
    constexpr int NST = (((Q4_G * LENF) >> 2) + 63) / 64;                  // 16-byte wave-stores per unit
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = lane >> 3, l = lane & 7;
    const int e = l < 4 ? l : 11 - l;
    const int k1 = ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1);
    float* const wstage = smem + w * STAGE;
    cf* const tabs = reinterpret_cast<cf*>(smem + Q4_WAVES * STAGE);
    cf* const winl = tabs;                                                 // [8][Q4_ROW] window pairs of samples e + 8m
    cf* const w200l = tabs + 8 * Q4_ROW;
    cf* const w400l = tabs + 16 * Q4_ROW;
    unsigned* const next_unit = reinterpret_cast<unsigned*>(tabs + 24 * Q4_ROW);
    int* const mlo = reinterpret_cast<int*>(next_unit + 4);                // MEL: first bins [slot][lane]; the weights
    float* const mwl = reinterpret_cast<float*>(mlo + lm_desc_ints(8));
    if constexpr (MEL) lane_mel_load_tables<S, 8, Q4_FLY>(mlo, mwl, mel, threadIdx.x, Q4_WAVES * 64);
    for (int i = threadIdx.x; i < 8 * Q4_ROW; i += Q4_WAVES * 64) {
        const int ll = i / Q4_ROW, m = i - ll * Q4_ROW;
        const int ee = ll < 4 ? ll : 11 - ll;
        winl[i] = m < Q4_M ? window_pair(g, ee + 8 * m) : mkc(0.0f, 0.0f);
        w200l[i] = tb.w200[i];
        w400l[i] = tb.w400[i];
    }

    // stage constants of the cross-lane 8-point transform: r = (partner + s * mine) * c
    const float s1 = e >= 4 ? -1.0f : 1.0f, s2 = (e & 2) ? -1.0f : 1.0f, s3 = (e & 1) ? -1.0f : 1.0f;
    const float R = 0.70710678118654752f;
    cf c1 = mkc(1.0f, 0.0f), c2 = mkc(1.0f, 0.0f);
    if (e == 5) c1 = mkc(R, -R);
    if (e == 6) c1 = mkc(0.0f, -1.0f);
    if (e == 7) c1 = mkc(-R, -R);
    if ((e & 3) == 3) c2 = mkc(0.0f, -1.0f);
    // source lane (byte address for ds_bpermute) of the k2 = 0 partner Z[25 * ((8 - k1) & 7)]
    int p0lane = l;
    if (l == 2) p0lane = 3;
    if (l == 3) p0lane = 2;
    if (l == 4) p0lane = 7;
    if (l == 7) p0lane = 4;
    if (l == 5) p0lane = 6;
    if (l == 6) p0lane = 5;
    const int p0addr = ((lane & ~7) | p0lane) << 2;

    const int T = (int)g.n_frames;
    const int upr = (T + Q4_G - 1) / Q4_G;                                 // units per row
    const int total = (int)g.rows * upr;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total ? begin + chunk : total;
    const float hscale = 0.5f * g.scale;                                   // the R2C split returns 2 X
    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + Q4_WAVES);
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };

    cf raw[Q4_M];
    // every lane group requests its own frame; a unit takes the fast path only if ALL its frames are interior
    auto prefetch = [&](int unit) -> bool {
        const int urow = unit / upr;
        const int frame = (unit - urow * upr) * Q4_G + slot;
        const long long start = (long long)frame * g.hop - g.center_pad;
        const bool ok = g.vec2_ok && frame < T && start >= 0 && start + 400 <= g.length;
        const bool all_ok = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        if (all_ok) {
            const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)urow * g.row_stride + start) + e;
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) raw[m] = src[8 * m];
        }
        return all_ok;
    };
    bool pre = false;
    int unit = begin + w;
    if (unit < end) pre = prefetch(unit);
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0): the loop is entered with nothing in flight
    __syncthreads();

    while (unit < end) {
        const int nxt = grab();
        const int urow = unit / upr;
        const int uframe0 = (unit - urow * upr) * Q4_G;
        cf v[Q4_M];
        {
            cf wn[Q4_M];
            q4_read_row(winl + l * Q4_ROW, wn);
            if (pre) {
#pragma unroll
                for (int m = 0; m < Q4_M; ++m) v[m] = cmul_elem(raw[m], wn[m]);
            } else {
                // frames touching the padding / past the end of the row: sample by sample (rolled loop)
                const int frame = uframe0 + slot;
                const float* rp = g.wave + (long long)urow * g.row_stride;
                const int s0 = (int)((long long)frame * g.hop - g.center_pad);
                const int L = (int)g.length;
                const bool live = frame < T;
#pragma unroll 1
                for (int m = 0; m < Q4_M; ++m) {
                    bool z0, z1;
                    const int j0 = padded_index(s0 + 2 * (e + 8 * m), L, g.pad_mode, &z0);
                    const int j1 = padded_index(s0 + 2 * (e + 8 * m) + 1, L, g.pad_mode, &z1);
                    const float a0 = rp[j0], a1 = rp[j1];
                    const cf wv = winl[l * Q4_ROW + m];
                    // (registers are indexed at compile time only: the values travel through the unit's staging area)
                    reinterpret_cast<cf*>(wstage)[lane * Q4_M + m] =
                        mkc((live && !z0) ? a0 * wv.x : 0.0f, (live && !z1) ? a1 * wv.y : 0.0f);
                }
                wave_lds_fence();
#pragma unroll
                for (int m = 0; m < Q4_M; ++m) v[m] = reinterpret_cast<const cf*>(wstage)[lane * Q4_M + m];
                wave_lds_fence();
            }
        }
        q4_dft25(v);                                                       // (1)
        {
            cf tw[Q4_M];                                                   // (2)
            q4_read_row(w200l + l * Q4_ROW, tw);
#pragma unroll
            for (int k = 1; k < Q4_M; ++k) v[k] = cmul(v[k], tw[k]);
        }
        // request the next unit now: it lands while this one is transformed across lanes, split, staged and stored
        __builtin_amdgcn_sched_barrier(0);
        pre = false;
        if (nxt < end) pre = prefetch(nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < Q4_M; ++k) {                                   // (3)
            cf p = q4_dpp<Q4_HALF_MIRROR>(v[k]);
            v[k] = cmul(__builtin_elementwise_fma(v[k], mkc(s1, s1), p), c1);
            p = q4_dpp<Q4_QUAD_XOR2>(v[k]);
            v[k] = cmul(__builtin_elementwise_fma(v[k], mkc(s2, s2), p), c2);
            p = q4_dpp<Q4_QUAD_XOR1>(v[k]);
            v[k] = __builtin_elementwise_fma(v[k], mkc(s3, s3), p);
        }

        const long long g0 = ((long long)urow * T + uframe0) * (MEL ? mel.n_mels : LENF);
        const int a = MEL ? 0 : (int)(g0 & 3);
        float* const stage = wstage + a;                                   // LDS and global share their 16-byte phase
        float* const srow = stage + slot * (MEL ? Q4_MEL_PITCH : LENF);
        {
            cf tw[Q4_M];                                                   // (4)
            q4_read_row(w400l + l * Q4_ROW, tw);
            const cf z0p = mkc(__int_as_float(__builtin_amdgcn_ds_bpermute(p0addr, __float_as_int(v[0].x))),
                               __int_as_float(__builtin_amdgcn_ds_bpermute(p0addr, __float_as_int(v[0].y))));
            cf zp[Q4_M];
            zp[0] = z0p;
#pragma unroll
            for (int k = 1; k < Q4_M; ++k) zp[k] = q4_dpp<Q4_QUAD_XOR3>(q4_dpp<Q4_HALF_MIRROR>(v[Q4_M - k]));   // lane l ^ 4
#pragma unroll
            for (int k = 0; k < Q4_M; ++k) {
                const int bin = 25 * k1 + k;
                const cf ev = cadd_conj(v[k], zp[k]), d = csub_conj(v[k], zp[k]);
                const cf twd = cmul_rot(tw[k], d);
                if constexpr (MODE == 0) {
                    reinterpret_cast<cf*>(srow)[bin] = cscale(cadd(ev, twd), hscale);
                    if (k == 0 && l == 0) reinterpret_cast<cf*>(srow)[200] = cscale(csub_then_conj(ev, twd), hscale);
                } else {
                    const cf pw = cscale(power_pair(ev, twd), hscale * hscale);       // (|X[k]|^2, |X[200 - k]|^2)
                    srow[bin] = spectral_row_value<MODE>(pw.x, ep);
                    if (k == 0 && l == 0) srow[200] = spectral_row_value<MODE>(pw.y, ep);
                }
            }
            wave_lds_fence();
        }
        const int nlive = (T - uframe0) < Q4_G ? (T - uframe0) : Q4_G;
        if constexpr (MEL) {
            // band-sparse contraction of the frame's row, dB, mel rows staged behind the power rows
            const int am = (int)(g0 & 3);
            float* const mstage = wstage + Q4_MEL_OFF + am;
            lane_mel_contract<S, 8, Q4_FLY>(srow, Q4_BINS, mlo, mwl, l, mel, mstage + slot * mel.n_mels);
            wave_lds_fence();
            lane_mel_store<(Q4_G * LM_MAX_MELS) / 256>(mstage, am, nlive * mel.n_mels, mel.out + g0, lane);
        } else {
        // the unit's live rows leave as 1 + NST + 1 unconditional nontemporal stores (lanes past the end repeat a neighbour)
        const int len = nlive * LENF;
        float* const gdst = ep.out + g0;
        const int npre = (4 - a) & 3;
        const int nchunks = (len - npre) >> 2;
        {
            const int hmax = (npre > 1 ? npre : 1) - 1;
            const int hi = lane < hmax ? lane : hmax;
            gdst[hi] = stage[hi];
        }
        const q4_f4* const s4 = reinterpret_cast<const q4_f4*>(stage + npre);
        q4_f4* const g4 = reinterpret_cast<q4_f4*>(gdst + npre);
        const int last = nchunks - 1;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = (lane + 64 * i) < last ? (lane + 64 * i) : last;
            __builtin_nontemporal_store(s4[c], g4 + c);
        }
        {
            const int r = len - npre - 4 * nchunks;
            const int rmax = (r > 1 ? r : 1) - 1;
            const int ti = len - 1 - (lane < rmax ? lane : rmax);
            gdst[ti] = stage[ti];
        }
        }
        wave_lds_fence();   // the next unit's staging writes must follow these reads
        unit = nxt;
    }
}

}  // namespace tac
#include "stft_n400_s3.hpp"
namespace tac {

// dynamic LDS of the kernel: staging areas, the three tables, the unit counter (the fused form adds lm_lds_bytes)
static size_t q4_lds_bytes(int) {
    return (size_t)Q4_WAVES * Q4_STAGE * sizeof(float) + (size_t)24 * Q4_ROW * sizeof(cf) + 16;
}

// device tables, one set per device
static int q4_tables(Q4Tables* out) {
    static std::mutex mu;
    static std::map<int, Q4Tables> cache;
    int dev = 0;
    TAC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) {
        *out = it->second;
        return TAC_OK;
    }
    std::vector<cf> host(3 * 8 * Q4_ROW, mkc(0.0f, 0.0f));
    const double two_pi = 6.283185307179586476925286766559;
    for (int l = 0; l < 8; ++l)
        for (int k2 = 0; k2 < Q4_M; ++k2) {
            const double a = -two_pi * (double)((q4_e_of(l) * k2) % 200) / 200.0;
            const double b = -two_pi * (double)(25 * q4_k1_of(l) + k2) / 400.0;
            host[l * Q4_ROW + k2] = mkc((float)std::cos(a), (float)std::sin(a));
            host[8 * Q4_ROW + l * Q4_ROW + k2] = mkc((float)std::cos(b), (float)std::sin(b));
            const double c = -two_pi * (double)(q4_e_of(l) + 8 * k2) / 400.0;
            host[16 * Q4_ROW + l * Q4_ROW + k2] = mkc((float)std::cos(c), (float)std::sin(c));
        }
    cf* dptr = nullptr;
    TAC_HIP(hipMalloc((void**)&dptr, host.size() * sizeof(cf)));
    TAC_HIP(hipMemcpy(dptr, host.data(), host.size() * sizeof(cf), hipMemcpyHostToDevice));
    Q4Tables t{dptr, dptr + 8 * Q4_ROW, dptr + 16 * Q4_ROW};
    cache[dev] = t;
    *out = t;
    return TAC_OK;
}

static bool q4_three_waves() {
    static const bool on = [] { const char* e = getenv("TAC_N400_TWO"); return !(e && e[0] == '1'); }();
    return on;
}

template <int MODE>
static int launch_n400(const FrameGeom& g, const Q4Tables& tb, const StftEpilogue& ep, hipStream_t stream) {
    const long long units = g.rows * ((g.n_frames + Q4_G - 1) / Q4_G);
    if (units >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const size_t bytes = q4_lds_bytes(0);
    long long blocks = (units + Q4_WAVES - 1) / Q4_WAVES;
    const long long cap = (long long)device_cu_count();
    if (blocks > cap) blocks = cap;
    // (the twelve-wave form, stft_n400_s3.hpp, serves the fused mel chain only: on these store-bound rows it measured 0-4 % slower)
    auto kern = stft_n400_kernel<MODE, false, 1>;
    if (bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Q4_WAVES * 64), bytes, stream, g, tb, ep, LaneMel{});
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

template <int MODE, int S>
static int launch_n400_mel_mode(const FrameGeom& g, const Q4Tables& tb, const LaneMel& mel, hipStream_t stream) {
    const long long units = g.rows * ((g.n_frames + Q4_G - 1) / Q4_G);
    if (units >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const size_t bytes = q4_lds_bytes(0) + lm_lds_bytes(8, mel.wtot);
    if (bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    long long blocks = (units + Q4_WAVES - 1) / Q4_WAVES;
    const long long cap = (long long)device_cu_count();
    if (blocks > cap) blocks = cap;
    if (q4_three_waves() && g.length >= 400) {
        const size_t b3 = q4s3_lds_bytes(MODE, true) + lm_lds_bytes(8, mel.wtot);
        if (b3 <= 160 * 1024) {
            long long bl = (units + Q4S3_WAVES - 1) / Q4S3_WAVES;
            if (bl > cap) bl = cap;
            auto k3 = stft_n400_s3_kernel<MODE, true, S>;
            TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(k3), (int)b3));
            hipLaunchKernelGGL(k3, dim3((unsigned)bl), dim3(Q4S3_WAVES * 64), b3, stream, g, tb,
                               StftEpilogue{nullptr, 1, 1, MODE == 1 ? 2.0f : 1.0f, 0, 0.0f, 0.0f}, mel, (const void*)nullptr,
                               (const float*)nullptr);
            TAC_HIP(hipGetLastError());
            return TAC_OK;
        }
    }
    auto kern = stft_n400_kernel<MODE, true, S>;
    if (bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Q4_WAVES * 64), bytes, stream, g, tb,
                       StftEpilogue{nullptr, 1, 1, MODE == 1 ? 2.0f : 1.0f, 0, 0.0f, 0.0f}, mel);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// int16 PCM / mu-law codes read by the fused kernel itself (power 2, twelve-wave kernel only)
template <int S, int FMT>
static int launch_n400_mel_coded(FrameGeom g, const Q4Tables& tb, const LaneMel& mel, hipStream_t stream, const void* samples,
                                 const float* lut) {
    const long long units = g.rows * ((g.n_frames + Q4_G - 1) / Q4_G);
    if (units >= 0x7fffffffLL || !q4_three_waves() || g.length < 400) return TAC_E_UNSUPPORTED;
    const size_t b3 = q4s3_lds_bytes(1, true) + lm_lds_bytes(8, mel.wtot) + 1024;
    if (b3 > 160 * 1024) return TAC_E_UNSUPPORTED;
    {                                                                      // sample pairs fetched as one access of the format
        const uintptr_t pair = FMT == FMT_I16 ? 4 : (FMT == FMT_MULAW_U8 ? 2 : 8);
        g.vec2_ok = ((g.hop & 1) == 0) && ((g.center_pad & 1) == 0) && ((g.row_stride & 1) == 0) &&
                    ((reinterpret_cast<uintptr_t>(samples) & (pair - 1)) == 0);
    }
    long long bl = (units + Q4S3_WAVES - 1) / Q4S3_WAVES;
    if (bl > device_cu_count()) bl = device_cu_count();
    auto k3 = stft_n400_s3_kernel<1, true, S, FMT>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(k3), (int)b3));
    hipLaunchKernelGGL(k3, dim3((unsigned)bl), dim3(Q4S3_WAVES * 64), b3, stream, g, tb, StftEpilogue{nullptr, 1, 1, 2.0f, 0, 0.0f, 0.0f},
                       mel, samples, lut);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

template <int FMT>
static int launch_n400_mel_coded_s(const FrameGeom& g, const Q4Tables& tb, const LaneMel& mel, int S, hipStream_t stream,
                                   const void* samples, const float* lut) {
    switch (S) {
#define TAC_Q4_CASE(SS) case SS: return launch_n400_mel_coded<SS, FMT>(g, tb, mel, stream, samples, lut);
        TAC_Q4_CASE(1) TAC_Q4_CASE(2) TAC_Q4_CASE(3) TAC_Q4_CASE(4) TAC_Q4_CASE(5) TAC_Q4_CASE(6) TAC_Q4_CASE(7) TAC_Q4_CASE(8)
        TAC_Q4_CASE(9) TAC_Q4_CASE(10) TAC_Q4_CASE(11) TAC_Q4_CASE(12)
#undef TAC_Q4_CASE
        default: return TAC_E_INVALID;
    }
}

// The fused Melspectrogram (+dB) chain for fft_length 400 (melspec_sparse.hip's entry points call these).
int launch_n400_mel(const FrameGeom& g, float power, const float* wpack, const int* desc, const int32_t* info_host,
                    int n_mels, int db, float amin, float log10_ref, float* out, hipStream_t stream, int fmt,
                    const void* samples, const float* lut) {
    if (!lane_mel_info_ok(info_host, 8, Q4_FLY)) return TAC_E_INVALID;
    if (n_mels < LM_MIN_MELS || n_mels > LM_MAX_MELS) return TAC_E_UNSUPPORTED;
    Q4Tables tb;
    const int rc = q4_tables(&tb);
    if (rc != TAC_OK) return rc;
    const LaneMel mel{wpack, desc, info_host[1], info_host[0], n_mels, db, amin, log10_ref, out, info_host[5] ? 1 : 0};
    if (fmt != FMT_F32) {
        if (power != 2.0f) return TAC_E_UNSUPPORTED;
        switch (fmt) {
            case FMT_I16: return launch_n400_mel_coded_s<FMT_I16>(g, tb, mel, info_host[4], stream, samples, lut);
            case FMT_MULAW_U8: return launch_n400_mel_coded_s<FMT_MULAW_U8>(g, tb, mel, info_host[4], stream, samples, lut);
            default: return launch_n400_mel_coded_s<FMT_MULAW_I64>(g, tb, mel, info_host[4], stream, samples, lut);
        }
    }
    const bool p2 = power == 2.0f;
    switch (info_host[4]) {                                                // steps per band
#define TAC_Q4_CASE(SS) case SS: return p2 ? launch_n400_mel_mode<1, SS>(g, tb, mel, stream) : launch_n400_mel_mode<2, SS>(g, tb, mel, stream);
        TAC_Q4_CASE(1) TAC_Q4_CASE(2) TAC_Q4_CASE(3) TAC_Q4_CASE(4) TAC_Q4_CASE(5) TAC_Q4_CASE(6) TAC_Q4_CASE(7) TAC_Q4_CASE(8)
        TAC_Q4_CASE(9) TAC_Q4_CASE(10) TAC_Q4_CASE(11) TAC_Q4_CASE(12)
#undef TAC_Q4_CASE
        default: return TAC_E_INVALID;
    }
}

// Band-slot layout of the fused form (mel_lanes.hpp).  h: the (n_freqs x n_mels) bank on the host.
int pack_n400(const std::vector<float>& h, int n_freqs, int n_mels, float* wpack, int wpack_cap, int32_t* desc,
              int desc_cap, int32_t* info_host, hipStream_t stream, bool to_host) {
    if (n_freqs != Q4_BINS) return TAC_E_UNSUPPORTED;
    return pack_lane_mel(h, n_freqs, n_mels, 8, Q4_MEL_PITCH, 1, Q4_FLY, LM_MAX_STEPS, q4_lds_bytes(0), wpack, wpack_cap, desc, desc_cap, info_host,
                         stream, to_host);
}

// ---------------------------------------------------------------- gradient: the inverse real transform per frame
// grad_frames[row][t][n] = window[n] * scale * Re sum_{k=0}^{200} G[k] e^{+2 pi i k n / 400}  (tac_amd.h (9)) for
// fft_length 400, on the same mixed-radix core run on conjugated data (the scheme of backward.hip's power-of-two kernel):
//   H[k] = G[k] (0 < k < 200), H[0] = 2 Re G[0], H[200] = 2 Re G[200];   conj Z[k] = c2r_operand(H[k], H[200 - k], W_400^k)
//   R = FFT_200(conj Z);   y[2m] = Re R[m], y[2m + 1] = -Im R[m], times window / 2.
// The core takes its input index p = e + 8 m in lane e, register m — here the FREQUENCY index k — and leaves output index
// 25 k1 + k2 in lane k1, register k2 — here the sample pair m: every lane ends with fifty consecutive samples of its frame.
// A unit is eight consecutive frames of one row: their gradient rows (SRC_NORM: spectrum rows + rows of the gradient of
// |z|^power, the norm's adjoint formed on the way in) are adjacent in memory and are parked in the wave's staging area,
// which then takes the unit's eight frame gradients and leaves as 16-byte stores.
// MELADJ (SRC_WAVE only): `gnorm` is the gradient of the MEL values, (rows, T, n_mels) frame-major, and the filterbank adjoint
// (two multiply-adds per bin through the 201-entry table `adj`, functional.py:183-184 transposed) is formed where the
// gradient of |X|^p is needed: the 4 F bytes per frame that fb_adjoint_kernel writes and this kernel reads never exist.
// OLA (SRC_WAVE only; hop and centre padding multiples of 4, hop >= 50): the unit IS a segment of ola_plan.hpp — its eight
// frame gradients are overlap-added out of the staging area: complete positions of clean frames go straight into the
// waveform gradient, the segment's border zone to gpad, its tail to edge[row][segment] (ola_fold_kernel finishes those and
// the padding images) — so neither the 1600 bytes of frame gradient per frame nor the gather kernel's pass over them exist.
template <int SRC, bool POW2, bool MELADJ = false, bool OLA = false>
__global__ void __launch_bounds__(Q4_WAVES * 64, 2)
stft_n400_backward_kernel(FrameGeom g, Q4Tables tb, const float* __restrict__ gspec, const float* __restrict__ gnorm,
                          float power, float* __restrict__ frames, const AdjEntry* __restrict__ adj, int n_mels,
                          float* __restrict__ gpad, float* __restrict__ edge, OlaPlan plan) {
    static_assert(!MELADJ || SRC == SRC_WAVE, "the fused adjoint re-transforms the frames");
    static_assert(!OLA || SRC == SRC_WAVE, "the in-kernel overlap-add serves the Spectrogram / Melspectrogram adjoints");
    constexpr int STAGE = Q4_STAGE;
    constexpr int ROWC = Q4_BINS;                                          // complex per staged gradient row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = lane >> 3, l = lane & 7;
    const int e = l < 4 ? l : 11 - l;
    const int k1 = ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1);
    cf* const wstage = reinterpret_cast<cf*>(smem + w * STAGE);
    cf* const tabs = reinterpret_cast<cf*>(smem + Q4_WAVES * STAGE);
    cf* const winl = tabs;                                                 // [8][Q4_ROW] window pairs of samples 2 (25 k1 + k2), + 1
    cf* const w200l = tabs + 8 * Q4_ROW;
    cf* const w400l = tabs + 16 * Q4_ROW;                                  // input layout: W_400^{e + 8 m}
    cf* const winf = tabs + 24 * Q4_ROW;                                   // SRC_WAVE: the forward transform's window pairs (input layout)
    cf* const w400o = tabs + 32 * Q4_ROW;                                  // SRC_WAVE: its R2C twiddles (output layout)
    unsigned* const next_unit = reinterpret_cast<unsigned*>(tabs + 40 * Q4_ROW);
    AdjEntry* const adj_lds = reinterpret_cast<AdjEntry*>(next_unit + 4);   // MELADJ: the per-bin table, then 8 mel-gradient rows per wave
    float* const grow = reinterpret_cast<float*>(adj_lds + Q4_BINS) + (w * Q4_G + slot) * 128;
    if constexpr (MELADJ)
        for (int i = threadIdx.x; i < Q4_BINS; i += Q4_WAVES * 64) adj_lds[i] = adj[i];
    const float wscale = 0.5f * g.scale;
    for (int i = threadIdx.x; i < 8 * Q4_ROW; i += Q4_WAVES * 64) {
        const int ll = i / Q4_ROW, m = i - ll * Q4_ROW;
        const cf wn = m < Q4_M ? window_pair(g, 25 * q4_k1_of(ll) + m) : mkc(0.0f, 0.0f);
        winl[i] = mkc(wn.x * wscale, -wn.y * wscale);                      // (y[2m + 1] = -Im R[m])
        w200l[i] = tb.w200[i];
        w400l[i] = tb.w400b[i];
        if constexpr (SRC == SRC_WAVE) {
            winf[i] = m < Q4_M ? window_pair(g, q4_e_of(ll) + 8 * m) : mkc(0.0f, 0.0f);
            w400o[i] = tb.w400[i];
        }
    }
    const float s1 = e >= 4 ? -1.0f : 1.0f, s2 = (e & 2) ? -1.0f : 1.0f, s3 = (e & 1) ? -1.0f : 1.0f;
    const float R = 0.70710678118654752f;
    cf c1 = mkc(1.0f, 0.0f), c2 = mkc(1.0f, 0.0f);
    if (e == 5) c1 = mkc(R, -R);
    if (e == 6) c1 = mkc(0.0f, -1.0f);
    if (e == 7) c1 = mkc(-R, -R);
    if ((e & 3) == 3) c2 = mkc(0.0f, -1.0f);
    int p0lane = l;                                        // (SRC_WAVE) source lane of the k2 = 0 partner, as in stft_n400_kernel
    if (l == 2) p0lane = 3;
    if (l == 3) p0lane = 2;
    if (l == 4) p0lane = 7;
    if (l == 7) p0lane = 4;
    if (l == 5) p0lane = 6;
    if (l == 6) p0lane = 5;
    const int p0addr = ((lane & ~7) | p0lane) << 2;

    const int T = (int)g.n_frames;
    const int upr = (T + Q4_G - 1) / Q4_G;
    const int total = (int)g.rows * upr;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total ? begin + chunk : total;
    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + Q4_WAVES);
    __syncthreads();
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    constexpr int NLD = (Q4_G * ROWC + 63) / 64;                           // complex values a lane parks per unit
    int unit = begin + w;
    while (unit < end) {
        const int nxt = grab();
        const int urow = unit / upr;
        const int uframe0 = (unit - urow * upr) * Q4_G;
        const int nlive = (T - uframe0) < Q4_G ? (T - uframe0) : Q4_G;
        const long long f0 = (long long)urow * T + uframe0;                // first frame of the unit (global frame index)
        // (1) the unit's gradient rows -> staging area (rows past the end of the waveform's frames: zeros)
        if constexpr (SRC == SRC_WAVE) {
            // no spectrum in memory: the unit's frames are fetched from the WAVEFORM and transformed by the forward
            // recipe of stft_n400_kernel; the R2C split leaves X[25 k1 + k2] in lane k1, where the gradient of |X|^p
            // (25 contiguous floats per lane) turns it into the gradient spectrum that is parked for step (2)
            const int frame = uframe0 + slot;
            const bool live = frame < T;
            float gv[MELADJ ? 16 : Q4_M], g200 = 0.0f;
            if constexpr (MELADJ) {                                         // the frame's mel-gradient row: 16 values per lane of its eight
                const float* gn = gnorm + (f0 + (live ? slot : 0)) * n_mels;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int b = l + 8 * q;
                    gv[q] = gn[b < n_mels ? b : n_mels - 1];
                }
            } else {
                const float* gn = gnorm + (f0 + (live ? slot : 0)) * ROWC + 25 * k1;
                g200 = gn[200 - 25 * k1];
#pragma unroll
                for (int k = 0; k < Q4_M; ++k) gv[k] = gn[k];
            }
            const long long start = (long long)frame * g.hop - g.center_pad;
            const bool ok = g.vec2_ok && live && start >= 0 && start + 400 <= g.length;
            const bool all_ok = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
            cf v[Q4_M];
            {
                cf wn[Q4_M];
                q4_read_row(winf + l * Q4_ROW, wn);
                if (all_ok) {
                    const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)urow * g.row_stride + start) + e;
#pragma unroll
                    for (int m = 0; m < Q4_M; ++m) v[m] = cmul_elem(src[8 * m], wn[m]);
                } else {
                    const float* rp = g.wave + (long long)urow * g.row_stride;
                    const int s0 = (int)start, L = (int)g.length;
#pragma unroll 1
                    for (int m = 0; m < Q4_M; ++m) {
                        bool z0, z1;
                        const int j0 = padded_index(s0 + 2 * (e + 8 * m), L, g.pad_mode, &z0);
                        const int j1 = padded_index(s0 + 2 * (e + 8 * m) + 1, L, g.pad_mode, &z1);
                        const float a0 = rp[j0], a1 = rp[j1];
                        const cf wv = winf[l * Q4_ROW + m];
                        wstage[lane * Q4_M + m] = mkc((live && !z0) ? a0 * wv.x : 0.0f, (live && !z1) ? a1 * wv.y : 0.0f);
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int m = 0; m < Q4_M; ++m) v[m] = wstage[lane * Q4_M + m];
                    wave_lds_fence();
                }
            }
            q4_dft25(v);
            {
                cf tw[Q4_M];
                q4_read_row(w200l + l * Q4_ROW, tw);
#pragma unroll
                for (int k = 1; k < Q4_M; ++k) v[k] = cmul(v[k], tw[k]);
            }
#pragma unroll
            for (int k = 0; k < Q4_M; ++k) {
                cf p = q4_dpp<Q4_HALF_MIRROR>(v[k]);
                v[k] = cmul(__builtin_elementwise_fma(v[k], mkc(s1, s1), p), c1);
                p = q4_dpp<Q4_QUAD_XOR2>(v[k]);
                v[k] = cmul(__builtin_elementwise_fma(v[k], mkc(s2, s2), p), c2);
                p = q4_dpp<Q4_QUAD_XOR1>(v[k]);
                v[k] = __builtin_elementwise_fma(v[k], mkc(s3, s3), p);
            }
            {
                cf tw[Q4_M];
                q4_read_row(w400o + l * Q4_ROW, tw);
                const cf z0p = mkc(__int_as_float(__builtin_amdgcn_ds_bpermute(p0addr, __float_as_int(v[0].x))),
                                   __int_as_float(__builtin_amdgcn_ds_bpermute(p0addr, __float_as_int(v[0].y))));
                const float hscale = 0.5f * g.scale;
                cf* row = wstage + slot * ROWC;
                if constexpr (MELADJ) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (l + 8 * q < n_mels) grow[l + 8 * q] = live ? gv[q] : 0.0f;
                    wave_lds_fence();
                }
                auto bin_grad = [&](int bin) {                              // (grad_mel . fb^T)[bin]
                    const AdjEntry en = adj_lds[bin];
                    return __builtin_fmaf(en.w0, grow[en.b0], en.w1 * grow[en.b1]);
                };
#pragma unroll
                for (int k = 0; k < Q4_M; ++k) {
                    const cf zp = k == 0 ? z0p : q4_dpp<Q4_QUAD_XOR3>(q4_dpp<Q4_HALF_MIRROR>(v[Q4_M - k]));   // lane l ^ 4
                    const cf ev = cadd_conj(v[k], zp), d = csub_conj(v[k], zp);
                    const cf twd = cmul_rot(tw[k], d);
                    float gvk;
                    if constexpr (MELADJ) gvk = bin_grad(25 * k1 + k);
                    else gvk = live ? gv[k] : 0.0f;
                    cf gk = norm_pow_grad<POW2>(cscale(cadd(ev, twd), hscale), gvk, power);
                    const int bin = 25 * k1 + k;
                    if (bin == 0) gk = mkc(2.0f * gk.x, 0.0f);                  // H[0] = 2 Re G[0]
                    row[bin] = gk;
                    if (k == 0 && l == 0) {
                        float g2v;
                        if constexpr (MELADJ) g2v = bin_grad(200);
                        else g2v = live ? g200 : 0.0f;
                        const cf g2 = norm_pow_grad<POW2>(cscale(csub_then_conj(ev, twd), hscale), g2v, power);
                        row[200] = mkc(2.0f * g2.x, 0.0f);                      // H[200] = 2 Re G[200]
                    }
                }
            }
        } else {
            const cf* src = reinterpret_cast<const cf*>(gspec) + f0 * ROWC;
            const float* gn = (SRC == SRC_NORM) ? gnorm + f0 * ROWC : nullptr;
            const int live = nlive * ROWC;
            cf val[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = lane + 64 * i;
                const int c = idx < live ? idx : live - 1;
                val[i] = src[c];
                if constexpr (SRC == SRC_NORM) val[i] = norm_pow_grad<POW2>(val[i], gn[c], power);
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = lane + 64 * i;
                if (idx < Q4_G * ROWC) {
                    cf x = idx < live ? val[i] : mkc(0.0f, 0.0f);
                    const int k = idx % ROWC;
                    if (k == 0 || k == 200) x = mkc(2.0f * x.x, 0.0f);      // H[0] = 2 Re G[0], H[200] = 2 Re G[200]
                    wstage[idx] = x;
                }
            }
        }
        wave_lds_fence();
        // (2) operands of the inverse transform: conj Z[e + 8 m]
        cf v[Q4_M];
        {
            cf tw[Q4_M];
            q4_read_row(w400l + l * Q4_ROW, tw);
            const cf* row = wstage + slot * ROWC;
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) {
                const cf hk = row[e + 8 * m], hm = row[200 - e - 8 * m];
                v[m] = c2r_operand(hk, hm, tw[m]);
            }
        }
        q4_dft25(v);
        {
            cf tw[Q4_M];
            q4_read_row(w200l + l * Q4_ROW, tw);
#pragma unroll
            for (int k = 1; k < Q4_M; ++k) v[k] = cmul(v[k], tw[k]);
        }
#pragma unroll
        for (int k = 0; k < Q4_M; ++k) {
            cf p = q4_dpp<Q4_HALF_MIRROR>(v[k]);
            v[k] = cmul(__builtin_elementwise_fma(v[k], mkc(s1, s1), p), c1);
            p = q4_dpp<Q4_QUAD_XOR2>(v[k]);
            v[k] = cmul(__builtin_elementwise_fma(v[k], mkc(s2, s2), p), c2);
            p = q4_dpp<Q4_QUAD_XOR1>(v[k]);
            v[k] = __builtin_elementwise_fma(v[k], mkc(s3, s3), p);
        }
        wave_lds_fence();                                                  // every lane's row reads precede the frame writes
        // (3) windowed sample pairs 25 k1 + k2 of the frame -> staging area (frame-major: 200 pairs per frame)
        {
            cf wn[Q4_M];
            q4_read_row(winl + l * Q4_ROW, wn);
            cf* fr = wstage + slot * 200 + 25 * k1;
#pragma unroll
            for (int k = 0; k < Q4_M; ++k) fr[k] = cmul_elem(v[k], wn[k]);
        }
        wave_lds_fence();
        if constexpr (OLA) {
            // (4') overlap-add of the unit's frames: a lane owns four consecutive positions o .. o + 3 of the unit's span
            // (hop, 400 and the padding are multiples of four: every frame covers all four or none, at a 16-byte offset)
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const q4_f4* s4 = reinterpret_cast<const q4_f4*>(wstage);
            const int hop = g.hop, seg = unit - urow * upr, seg_span = Q4_G * hop, open = 400 - hop;
            const int span = (nlive - 1) * hop + 400;
            const int p0 = seg * seg_span;                                  // padded position of the unit's first sample
            float* const prow = gpad + (long long)urow * plan.pad_len;
            float* const erow = edge + ((long long)urow * (plan.segs_per_row - 1) + seg) * open;
            float* const orow = plan.gwave + (long long)urow * plan.gstride;
            const float inv_hop = 1.0f / (float)hop;
            for (int o = 4 * lane; o < span; o += 256) {
                q4_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < Q4_G; ++q) {
                    const int n = o - q * hop;
                    if (q < nlive && n >= 0 && n < 400) acc += s4[q * 100 + (n >> 2)];
                }
                f4u out4;
                out4.x = acc.x; out4.y = acc.y; out4.z = acc.z; out4.w = acc.w;
                if (o < seg_span) {
                    const int qf = (int)(((float)o + 0.5f) * inv_hop);      // o / hop (exact: o < 2^13)
                    const int p = p0 + o;
                    const int fc = seg * Q4_G + qf;                         // (the fold kernel skips exactly these runs)
                    if (fc < T && ola_direct(g, plan, fc)) *reinterpret_cast<f4u*>(orow + (p - g.center_pad)) = out4;
                    else *reinterpret_cast<f4u*>(prow + p) = out4;
                } else if (seg < plan.segs_per_row - 1) {
                    *reinterpret_cast<f4u*>(erow + (o - seg_span)) = out4;
                } else {
                    *reinterpret_cast<f4u*>(prow + p0 + o) = out4;
                }
            }
        } else {
        // (4) the live frames leave as 16-byte stores (a frame is 1600 bytes: the unit's run is 16-byte aligned)
        {
            const q4_f4* s4 = reinterpret_cast<const q4_f4*>(wstage);
            q4_f4* g4 = reinterpret_cast<q4_f4*>(frames + f0 * 400);
            const int nch = nlive * 100;
#pragma unroll
            for (int i = 0; i < (Q4_G * 100 + 63) / 64; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) g4[c] = s4[c];
            }
        }
        }
        wave_lds_fence();
        unit = nxt;
    }
}

// tac_stft_backward_f32 / tac_stft_norm_backward_f32 for fft_length 400 (backward.hip's dispatcher calls this)
int launch_n400_backward(const FrameGeom& g, const float* gspec, const float* gnorm, float power, float* frames,
                         hipStream_t stream, bool from_wave, const AdjEntry* adj, int n_mels, float* gpad, float* edge,
                         const OlaPlan* plan) {
    Q4Tables tb;
    const int rc = q4_tables(&tb);
    if (rc != TAC_OK) return rc;
    if ((!from_wave && (reinterpret_cast<uintptr_t>(gspec) & 7u)) || (!plan && (reinterpret_cast<uintptr_t>(frames) & 15u))) return TAC_E_UNSUPPORTED;
    if (plan && (!from_wave || !gpad || !edge || plan->seg_frames != Q4_G || (g.hop & 3) || (g.center_pad & 3) || g.hop < 50 ||
                 g.hop > 400 || !plan->gwave))
        return TAC_E_UNSUPPORTED;
    const long long units = g.rows * ((g.n_frames + Q4_G - 1) / Q4_G);
    if (units >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    if (adj && (!from_wave || n_mels < 1 || n_mels > 128)) return TAC_E_UNSUPPORTED;
    const size_t bytes = q4_lds_bytes(0) + (size_t)16 * Q4_ROW * sizeof(cf) +
                         (adj ? (size_t)Q4_BINS * sizeof(AdjEntry) + (size_t)Q4_WAVES * Q4_G * 128 * sizeof(float) : 0);
    long long blocks = (units + Q4_WAVES - 1) / Q4_WAVES;
    const long long cap = (long long)device_cu_count();
    if (blocks > cap) blocks = cap;
    void (*kern)(FrameGeom, Q4Tables, const float*, const float*, float, float*, const AdjEntry*, int, float*, float*, OlaPlan);
    if (from_wave && plan) {
        if (!gnorm) return TAC_E_INVALID;
        if (adj) kern = power == 2.0f ? stft_n400_backward_kernel<SRC_WAVE, true, true, true> : stft_n400_backward_kernel<SRC_WAVE, false, true, true>;
        else kern = power == 2.0f ? stft_n400_backward_kernel<SRC_WAVE, true, false, true> : stft_n400_backward_kernel<SRC_WAVE, false, false, true>;
    } else if (from_wave) {
        if (!gnorm) return TAC_E_INVALID;
        if (adj) kern = power == 2.0f ? stft_n400_backward_kernel<SRC_WAVE, true, true> : stft_n400_backward_kernel<SRC_WAVE, false, true>;
        else kern = power == 2.0f ? stft_n400_backward_kernel<SRC_WAVE, true> : stft_n400_backward_kernel<SRC_WAVE, false>;
    } else if (!gnorm) kern = stft_n400_backward_kernel<SRC_GRAD, false>;
    else kern = power == 2.0f ? stft_n400_backward_kernel<SRC_NORM, true> : stft_n400_backward_kernel<SRC_NORM, false>;
    if (bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Q4_WAVES * 64), bytes, stream, g, tb, gspec, gnorm, power, frames, adj,
                       n_mels, gpad, edge, plan ? *plan : OlaPlan{});
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// Entry used by stft_kernels.hip's dispatcher: TAC_E_UNSUPPORTED when this form does not apply (two-sided output,
// |X|^p with p outside {1, 2}); the caller then evaluates the DFT as a matrix product.
int try_launch_n400(const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream) {
    if (!ep.onesided) return TAC_E_UNSUPPORTED;
    int pmode = -1;
    if (mode == 0) pmode = 0;
    else if (ep.power == 2.0f) pmode = ep.db ? 3 : 1;
    else if (ep.power == 1.0f) pmode = ep.db ? 4 : 2;
    if (pmode < 0) return TAC_E_UNSUPPORTED;
    Q4Tables tb;
    const int rc = q4_tables(&tb);
    if (rc != TAC_OK) return rc;
    switch (pmode) {
        case 0: return launch_n400<0>(g, tb, ep, stream);
        case 1: return launch_n400<1>(g, tb, ep, stream);
        case 2: return launch_n400<2>(g, tb, ep, stream);
        case 3: return launch_n400<3>(g, tb, ep, stream);
        default: return launch_n400<4>(g, tb, ep, stream);
    }
}

}  // namespace tac
