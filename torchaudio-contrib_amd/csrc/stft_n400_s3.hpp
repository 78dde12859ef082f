// stft_n400_s3.hpp — stft_n400_kernel (stft_n400.hip) re-cut for THREE waves per SIMD: no second register set for a unit in
// flight (the next unit's samples are requested into the transform's own registers once the rows are staged, from an address
// clamped into the row), twelve <= 168-register waves per workgroup, a staging area sized for what the mode stages (the gather
// path of padded units goes through it in two halves).  Same arithmetic, same tables, same row epilogues.
// (included by stft_n400.hip behind the helpers it uses)
#pragma once

namespace tac {

constexpr int Q4S3_WAVES = 12;
// floats of LDS per wave: the unit's eight rows (+ the 16-byte phase) [+ the mel rows behind them], at least the 64 x 13 complex
// values of half a gathered unit
__host__ __device__ constexpr int q4s3_stage(int mode, bool mel) {
    const int rows = mel ? Q4_MEL_OFF + 4 + Q4_G * LM_MAX_MELS : (Q4_G * (mode == 0 ? 2 : 1) * Q4_BINS + 4);
    const int need = rows > 64 * 13 * 2 ? rows : 64 * 13 * 2;
    return ((need + 3) / 4) * 4;
}
inline size_t q4s3_lds_bytes(int mode, bool mel) {
    return (size_t)Q4S3_WAVES * q4s3_stage(mode, mel) * sizeof(float) + (size_t)24 * Q4_ROW * sizeof(cf) + 16;
}

// FMT (fused mel form): sample format of the frame load — int16 PCM (value = sample 2^-15, folded into the window table) and
// mu-law codes (uint8 / int64, 256-entry decode table in LDS) are converted in registers, like the other fused kernels do
template <int MODE, bool MEL, int S, int FMT = FMT_F32>
__global__ void __launch_bounds__(Q4S3_WAVES * 64, 3)
stft_n400_s3_kernel(FrameGeom g, Q4Tables tb, StftEpilogue ep, LaneMel mel, const void* __restrict__ samples = nullptr,
                    const float* __restrict__ lut = nullptr) {
    static_assert(FMT == FMT_F32 || MEL, "coded inputs: the fused Melspectrogram form");
    constexpr int LENF = (MODE == 0 ? 2 : 1) * Q4_BINS;
    constexpr int STAGE = q4s3_stage(MODE, MEL);                           // floats per wave
    constexpr int NST = (((Q4_G * LENF) >> 2) + 63) / 64;                  // 16-byte wave-stores per unit
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = lane >> 3, l = lane & 7;
    const int e = l < 4 ? l : 11 - l;
    const int k1 = ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1);
    float* const wstage = smem + w * STAGE;
    cf* const tabs = reinterpret_cast<cf*>(smem + Q4S3_WAVES * STAGE);
    cf* const winl = tabs;                                                 // [8][Q4_ROW] window pairs of samples e + 8m
    cf* const w200l = tabs + 8 * Q4_ROW;
    cf* const w400l = tabs + 16 * Q4_ROW;
    unsigned* const next_unit = reinterpret_cast<unsigned*>(tabs + 24 * Q4_ROW);
    int* const mlo = reinterpret_cast<int*>(next_unit + 4);                // MEL: first bins [slot][lane]; the weights
    float* const mwl = reinterpret_cast<float*>(mlo + lm_desc_ints(8));
    if constexpr (MEL) lane_mel_load_tables<S, 8, Q4_FLY>(mlo, mwl, mel, threadIdx.x, Q4S3_WAVES * 64);
    float* const lutlds = mwl + ((mel.wtot + 3) & ~3);                     // mu-law decode table behind the weights
    if constexpr (FMT >= FMT_MULAW_U8) {
        if (threadIdx.x < 256) lutlds[threadIdx.x] = lut[threadIdx.x];
    }
    const float pcm = FMT == FMT_I16 ? (1.0f / 32768.0f) : 1.0f;
    for (int i = threadIdx.x; i < 8 * Q4_ROW; i += Q4S3_WAVES * 64) {
        const int ll = i / Q4_ROW, m = i - ll * Q4_ROW;
        const int ee = ll < 4 ? ll : 11 - ll;
        winl[i] = m < Q4_M ? cscale(window_pair(g, ee + 8 * m), pcm) : mkc(0.0f, 0.0f);
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
    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + Q4S3_WAVES);
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };

    // every lane group requests its own frame, unconditionally, from an address clamped into the row, straight into the
    // transform's registers (no second set for a unit in flight: that is what fits twelve waves); a unit takes the fast
    // path only if ALL its frames are interior
    cf v[Q4_M];
    bool fast = false;
    auto request = [&](int unit) {
        unit = unit < end ? unit : end - 1;
        const int urow = unit / upr;
        const int frame = (unit - urow * upr) * Q4_G + slot;
        const long long start = (long long)frame * g.hop - g.center_pad;
        const bool ok = g.vec2_ok && frame < T && start >= 0 && start + 400 <= g.length;
        fast = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        long long cs = start < 0 ? 0 : start;
        cs = cs + 400 <= g.length ? cs : g.length - 400;
        const long long off = (long long)urow * g.row_stride + cs;           // in samples
        if constexpr (FMT == FMT_F32) {
            const cf* src = reinterpret_cast<const cf*>(g.wave + off) + e;
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) v[m] = src[8 * m];
        } else if constexpr (FMT == FMT_I16) {                              // a pair of samples = one dword
            const unsigned* src = reinterpret_cast<const unsigned*>(static_cast<const short*>(samples) + off) + e;
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) v[m].x = __uint_as_float(src[8 * m]);
        } else if constexpr (FMT == FMT_MULAW_U8) {                         // a pair of codes = one 16-bit load
            const unsigned short* src = reinterpret_cast<const unsigned short*>(static_cast<const unsigned char*>(samples) + off) + e;
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) v[m].x = __uint_as_float((unsigned)src[8 * m]);
        } else {                                                            // int64 codes: the low dword of each
            const int* src = reinterpret_cast<const int*>(static_cast<const long long*>(samples) + off) + 4 * e;
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) {
                v[m].x = __int_as_float(src[32 * m]);
                v[m].y = __int_as_float(src[32 * m + 2]);
            }
        }
    };
    // the requested registers as float sample pairs (still unwindowed): PCM integers / decoded codes
    auto decode = [&]() {
        if constexpr (FMT == FMT_I16) {
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) {
                const int bits = __float_as_int(v[m].x);
                v[m] = mkc((float)(short)(bits & 0xffff), (float)(bits >> 16));
            }
        } else if constexpr (FMT == FMT_MULAW_U8) {
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) {
                const unsigned bits = __float_as_uint(v[m].x);
                v[m] = mkc(lutlds[bits & 0xffu], lutlds[(bits >> 8) & 0xffu]);
            }
        } else if constexpr (FMT == FMT_MULAW_I64) {
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) v[m] = mkc(lutlds[__float_as_uint(v[m].x) & 0xffu], lutlds[__float_as_uint(v[m].y) & 0xffu]);
        }
    };
    auto sample_at = [&](long long row_offset, int j) -> float {            // the gather path's access, in the same units as `decode`
        if constexpr (FMT == FMT_F32) return g.wave[row_offset + j];
        else if constexpr (FMT == FMT_I16) return (float)static_cast<const short*>(samples)[row_offset + j];
        else if constexpr (FMT == FMT_MULAW_U8) return lutlds[static_cast<const unsigned char*>(samples)[row_offset + j]];
        else return lutlds[(unsigned)static_cast<const long long*>(samples)[row_offset + j] & 0xffu];
    };
    int unit = begin + w;
    __syncthreads();
    if (unit >= end) return;
    request(unit);

    while (unit < end) {
        const int nxt = grab();
        const int urow = unit / upr;
        const int uframe0 = (unit - urow * upr) * Q4_G;
        if (fast) {
            decode();
            cf wn[Q4_M];
            q4_read_row(winl + l * Q4_ROW, wn);
#pragma unroll
            for (int m = 0; m < Q4_M; ++m) v[m] = cmul_elem(v[m], wn[m]);
        } else {
            // frames touching the padding / past the end of the row: sample by sample (rolled loops), in two halves through
            // the (smaller) staging area — registers are indexed at compile time only
            const int frame = uframe0 + slot;
            const long long rbase = (long long)urow * g.row_stride;
            const int s0 = (int)((long long)frame * g.hop - g.center_pad);
            const int L = (int)g.length;
            const bool live = frame < T;
            constexpr int HALF0 = 13;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m0 = h * HALF0, m1 = h ? Q4_M : HALF0;
#pragma unroll 1
                for (int m = m0; m < m1; ++m) {
                    bool z0, z1;
                    const int j0 = padded_index(s0 + 2 * (e + 8 * m), L, g.pad_mode, &z0);
                    const int j1 = padded_index(s0 + 2 * (e + 8 * m) + 1, L, g.pad_mode, &z1);
                    const float a0 = sample_at(rbase, j0), a1 = sample_at(rbase, j1);
                    const cf wv = winl[l * Q4_ROW + m];
                    reinterpret_cast<cf*>(wstage)[lane * HALF0 + (m - m0)] =
                        mkc((live && !z0) ? a0 * wv.x : 0.0f, (live && !z1) ? a1 * wv.y : 0.0f);
                }
                wave_lds_fence();
#pragma unroll
                for (int m = 0; m < Q4_M; ++m)
                    if (m >= m0 && m < m1) v[m] = reinterpret_cast<const cf*>(wstage)[lane * HALF0 + (m - m0)];
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
        // the next unit's samples go out now (the transform's registers are free), before this unit's stores
        __builtin_amdgcn_sched_barrier(0);
        request(nxt);
        __builtin_amdgcn_sched_barrier(0);
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
