// stft_small3.hpp — stft_small_kernel (fft_length 512 / 1024, G = 4 / 2 frames per wave) re-cut for THREE or FOUR waves
// per SIMD, the recipe of melspec_stream3.hpp: nothing lane-dependent is hoisted for the kernel's lifetime — the pass-1
// twiddles, the R2C twiddles and the window are re-read from LDS every unit — and the next unit's samples are requested
// into the FFT's own registers once the unit's rows are staged, so a wave needs ~120 registers instead of ~250.
// The rows are staged in place over the unit's exchange areas, as before.  Replaces torch.stft (functional.py:36-38)
// [+ complex_norm (:58-72)] [+ apply_filterbank (:172-184)] [+ amplitude_to_db (:291-296)] at these sizes.
#pragma once
#include "host_common.hpp"
#include "mel_lanes.hpp"

namespace tac {

constexpr int SM3_TW_STRIDE = 36;     // floats between the 16 pass-1 twiddle sets (144 B: conflict-free 16-byte reads)
constexpr int SM3_TW_BYTES = 16 * SM3_TW_STRIDE * 4;

template <int NC>
inline size_t small3_lds_bytes(int waves) {
    using F = WaveFft<NC, 16>;
    constexpr int WAVE_SLOTS = ((F::G * F::PADDED + 1) / 2) * 2;
    return (size_t)waves * WAVE_SLOTS * sizeof(cf) + (size_t)F::LPF * 18 * sizeof(cf) + SM3_TW_BYTES +
           (size_t)4 * F::LPF * 16 + 16;
}

// FMT (fused mel form only): sample format of the frame load — int16 PCM (value = sample 2^-15, folded into the window) and mu-law
// codes (uint8 / int64; 256-entry decode table in LDS) are converted in registers, like melspec_stream3_kernel does at 2048
template <int NC, int MODE, bool MEL, int S, int WAVES, int FMT = FMT_F32>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4)
stft_small3_kernel(FrameGeom g, Tables tb, StftEpilogue ep, LaneMel mel, const void* __restrict__ samples = nullptr,
                   const float* __restrict__ lut = nullptr) {
    static_assert(FMT == FMT_F32 || MEL, "coded inputs: the fused Melspectrogram form");
    constexpr int E = 16;
    using F = WaveFft<NC, E>;
    constexpr int LPF = F::LPF, G = F::G, NPASS = F::NPASS;
    static_assert(G >= 2 && radix_at(NC, 0) == E && (NPASS == 2 || NPASS == 3), "fft_length 256 / 512 / 1024");
    // pass 1 uses ONE set of R1 - 1 twiddles per lane: a middle pass whose stride 16 divides LPF (row t & 15), or the last pass
    // (row t; its W_E^{bq} factors are compile-time constants inside pass_twiddle)
    constexpr int R1 = radix_at(NC, 1);
    static_assert(pass_shares_twiddles(NC, E, 1) && (LPF % 16 == 0 || pass_is_last(NC, 1)) && R1 <= 16, "one pass-1 twiddle set per lane");
    constexpr int TW1_ROWS = LPF < 16 ? LPF : 16;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / LPF, t = lane % LPF;
    constexpr int WAVE_SLOTS = ((G * F::PADDED + 1) / 2) * 2;
    cf* const wbase = smem + w * WAVE_SLOTS;
    cf* const lds = wbase + sub * F::PADDED;
    // window pairs, one 144-byte row per first-pass column (8 conflict-free ds_read_b128 per lane); the 1/2 of the R2C split
    // and the normalisation are folded in
    constexpr int WROW = E + 2;
    const float half = 0.5f * g.scale * (FMT == FMT_I16 ? (1.0f / 32768.0f) : 1.0f);
    cf* const wlds = smem + WAVES * WAVE_SLOTS;
    for (int m = threadIdx.x; m < NC; m += WAVES * 64) wlds[(m % LPF) * WROW + (m / LPF)] = cscale(window_pair(g, m), half);
    float* const twlds = reinterpret_cast<float*>(wlds + LPF * WROW);
    if (threadIdx.x < 16 * 16) {
        const int js = threadIdx.x >> 4, q = threadIdx.x & 15;
        if (js < TW1_ROWS && q < R1) {
            const cf wv = q ? tb.w_nc[js * q * (NC / (16 * R1))] : mkc(1.0f, 0.0f);
            twlds[js * SM3_TW_STRIDE + 2 * (q ? q - 1 : 15)] = wv.x;
            twlds[js * SM3_TW_STRIDE + 2 * (q ? q - 1 : 15) + 1] = wv.y;
        }
    }
    // the eight R2C twiddles of a column as [read u][column] 16-byte pairs
    cf* const ptwl = reinterpret_cast<cf*>(twlds + SM3_TW_BYTES / 4);
    for (int idx = threadIdx.x; idx < LPF * F::NPAIR; idx += WAVES * 64) {
        const int tt = idx / F::NPAIR, p = idx - tt * F::NPAIR;
        ptwl[((p >> 1) * LPF + tt) * 2 + (p & 1)] = tb.w_n[tt + p * LPF];
    }
    unsigned* const next_unit = reinterpret_cast<unsigned*>(ptwl + 2 * 4 * LPF);

    cf tw2 = mkc(1.0f, 0.0f);
    if constexpr (NPASS == 3) {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
        tw2 = all[twiddles_before(NC, E, 2)];
    }

    const int T = (int)g.n_frames;
    const int upr = (T + G - 1) / G;                       // units per row
    const int total = (int)g.rows * upr;
    // GORDER (round 5, complex rows): units dealt round-robin over all waves of the grid — local index u is unit (u / WAVES) G_W +
    // slot WAVES + u % WAVES (the slots of one XCD side by side) — so that the grid writes one tight window of adjacent rows instead
    // of one stream per workgroup: fft_length 512 / 1024 / 256 complex rows -8 / -6 / -6 % (process-level, alternating,
    // profiles/r05/ab/batch57); the real rows (half the bytes per unit) measure +-0 ... +2 % and keep their chunks.
    constexpr bool TAC_SM3_GORDER = MODE == 0 && !MEL;
    const int GW = (int)gridDim.x * WAVES;
    const int gslot = (gridDim.x & 7u) == 0 ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int nfull = total / GW, grem = total - nfull * GW - gslot * WAVES;
    const int chunk = TAC_SM3_GORDER ? nfull * WAVES + (grem < 0 ? 0 : (grem > WAVES ? WAVES : grem)) : (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = TAC_SM3_GORDER ? 0 : (int)blockIdx.x * chunk;
    const int end = TAC_SM3_GORDER ? chunk : (begin + chunk < total ? begin + chunk : total);
    auto unit_of = [&](int u) { return TAC_SM3_GORDER ? (u / WAVES) * GW + gslot * WAVES + u % WAVES : u; };
    constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);
    constexpr int NST = (((G * LENF) >> 2) + 63) / 64;    // 16-byte wave-stores per unit

    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + WAVES);
    constexpr int PITCH = sm_mel_pitch(NC), MEL_OFF = G * PITCH + 8;
    static_assert(!MEL || MEL_OFF + 4 + G * LM_MAX_MELS <= 2 * WAVE_SLOTS, "mel rows fit the wave's area");
    int* const mlo = reinterpret_cast<int*>(next_unit + 4);                // MEL: first bins [slot][lane]; the weights
    float* const mwl = reinterpret_cast<float*>(mlo + lm_desc_ints(LPF));
    if constexpr (MEL) lane_mel_load_tables<S, LPF, SM_FLY>(mlo, mwl, mel, threadIdx.x, WAVES * 64);
    float* const lutlds = mwl + ((mel.wtot + 3) & ~3);                     // mu-law decode table behind the weights
    if constexpr (FMT >= FMT_MULAW_U8) {
        if (threadIdx.x < 256) lutlds[threadIdx.x] = lut[threadIdx.x];
    }
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    // every lane group requests its own frame, unconditionally (start clamped into the row); the unit takes the fast path
    // only if ALL its frames are interior
    cf v[1][E];
    bool fast = false;
    auto request = [&](int unit) {
        unit = unit_of(unit < end ? unit : end - 1);
        const int urow = unit / upr;
        const int frame = (unit - urow * upr) * G + sub;
        const long long start = (long long)frame * g.hop - g.center_pad;
        const bool ok = g.vec2_ok && frame < T && start >= 0 && start + F::N <= g.length;
        fast = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        if (g.length < F::N) return;                                        // rows shorter than a frame: every unit gathers
        long long cs = start < 0 ? 0 : start;
        cs = cs + F::N <= g.length ? cs : g.length - F::N;
        const long long off = (long long)urow * g.row_stride + cs;           // in samples
        if constexpr (FMT == FMT_F32) {
            const cf* src = reinterpret_cast<const cf*>(g.wave + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[0][q] = src[t + q * LPF];
        } else if constexpr (FMT == FMT_I16) {                              // a pair of samples = one dword
            const unsigned* src = reinterpret_cast<const unsigned*>(static_cast<const short*>(samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[0][q].x = __uint_as_float(src[t + q * LPF]);
        } else if constexpr (FMT == FMT_MULAW_U8) {                         // a pair of codes = one 16-bit load
            const unsigned short* src = reinterpret_cast<const unsigned short*>(static_cast<const unsigned char*>(samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[0][q].x = __uint_as_float((unsigned)src[t + q * LPF]);
        } else {                                                            // int64 codes: the low dword of each
            const int* src = reinterpret_cast<const int*>(static_cast<const long long*>(samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                v[0][q].x = __int_as_float(src[4 * (t + q * LPF)]);
                v[0][q].y = __int_as_float(src[4 * (t + q * LPF) + 2]);
            }
        }
    };
    // the requested registers as float sample pairs (still unwindowed): PCM integers / decoded codes
    auto decode = [&]() {
        if constexpr (FMT == FMT_I16) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int bits = __float_as_int(v[0][q].x);
                v[0][q] = mkc((float)(short)(bits & 0xffff), (float)(bits >> 16));
            }
        } else if constexpr (FMT == FMT_MULAW_U8) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const unsigned bits = __float_as_uint(v[0][q].x);
                v[0][q] = mkc(lutlds[bits & 0xffu], lutlds[(bits >> 8) & 0xffu]);
            }
        } else if constexpr (FMT == FMT_MULAW_I64) {
#pragma unroll
            for (int q = 0; q < E; ++q)
                v[0][q] = mkc(lutlds[__float_as_uint(v[0][q].x) & 0xffu], lutlds[__float_as_uint(v[0][q].y) & 0xffu]);
        }
    };
    struct Fetch {                                        // sample access of the gather path, in the same units as `decode`
        const void* base;
        const float* lut;
        __device__ __forceinline__ float operator()(long long row_offset, int j) const {
            if constexpr (FMT == FMT_F32) return static_cast<const float*>(base)[row_offset + j];
            else if constexpr (FMT == FMT_I16) return (float)static_cast<const short*>(base)[row_offset + j];
            else if constexpr (FMT == FMT_MULAW_U8) return lut[static_cast<const unsigned char*>(base)[row_offset + j]];
            else return lut[(unsigned)static_cast<const long long*>(base)[row_offset + j] & 0xffu];
        }
    };
    int unit = begin + w;
    __syncthreads();
    if (unit >= end) return;
    request(unit);

    while (unit < end) {
        const int nxt = grab();
        const int urow = unit_of(unit) / upr;
        const int uframe0 = (unit_of(unit) - urow * upr) * G;
        // ---- window, pass 0 (units with frames in the padding or past the end of the row gather their samples first)
        if (!fast) {
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v[0], g, nullptr, lds, urow, uframe0 + sub, tz,
                                             Fetch{FMT == FMT_F32 ? static_cast<const void*>(g.wave) : samples, lutlds});
        } else {
            decode();
        }
        {
            const f4* wp = reinterpret_cast<const f4*>(wlds + t * WROW);
            cf win[E];
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                const f4 x = wp[i];
                win[2 * i] = mkc(x.x, x.y);
                win[2 * i + 1] = mkc(x.z, x.w);
            }
            Dft<16>::run_windowed(v[0], win);
        }
        wave_lds_fence();
        // ---- pass 1 (its fifteen twiddles from LDS) [and pass 2]; the last pass keeps the lower half of the spectrum in registers
        {
            cf tw1[16];
            const f4* tl = reinterpret_cast<const f4*>(twlds + (t & (TW1_ROWS - 1)) * SM3_TW_STRIDE);
#pragma unroll
            for (int u = 0; u < R1 / 2; ++u) {
                const f4 x = tl[u];
                tw1[2 * u] = mkc(x.x, x.y);
                tw1[2 * u + 1] = mkc(x.z, x.w);
            }
            F::template pass_write<0, true>(v[0], lds, t, t);
            wave_lds_fence();
            F::template pass_readback<1>(v[0], lds, t);
            F::template pass_twiddle<1, true>(v[0], tw1);
            F::template pass_butterflies<1>(v[0]);
            wave_lds_fence();
            F::template pass_write<1, true>(v[0], lds, t, t);
            wave_lds_fence();
            if constexpr (NPASS == 3) {
                F::template pass_readback<2>(v[0], lds, t);
                F::template pass_twiddle<2, true>(v[0], &tw2);
                F::template pass_butterflies<2>(v[0]);
                wave_lds_fence();
                F::template pass_write<2, true>(v[0], lds, t, t);
                wave_lds_fence();
            }
        }
        const long long g0 = ((long long)urow * T + uframe0) * (MEL ? mel.n_mels : LENF);
        const int a = MEL ? 0 : (int)(g0 & 3);
        float* const stage = reinterpret_cast<float*>(wbase) + a;        // LDS and global share their 16-byte phase
        float* const srow = stage + sub * (MEL ? PITCH : LENF);
        {
            cf ptw[F::NPAIR];
            const f4* pl = reinterpret_cast<const f4*>(ptwl) + t;
#pragma unroll
            for (int u = 0; u < F::NPAIR / 2; ++u) {
                const f4 x = pl[u * LPF];
                ptw[2 * u] = mkc(x.x, x.y);
                ptw[2 * u + 1] = mkc(x.z, x.w);
            }
            cf xa[F::NPAIR], xb[F::NPAIR], xm, unused;
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) {
                const int k = t + i * LPF;
                const cf zk = v[0][F::reg_of_spectrum(i)];
                const cf zm = (i == 0) ? F::r2c_partner(lds, k, zk) : lds[lds_pad(NC - k)];
                if constexpr (MODE != 0) xa[i] = F::r2c_power_x2(zk, zm, ptw[i]);   // (|X[k]|^2, |X[NC-k]|^2), no spectra formed
                else F::r2c_split_x2(zk, zm, ptw[i], xa[i], xb[i]);
            }
            F::r2c_pair(lds, NC / 2, mkc(0.0f, -1.0f), xm, unused);
            wave_lds_fence();                                             // every Z of this unit is in registers
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) {
                const int k = t + i * LPF;
                if constexpr (MODE == 0) {
                    reinterpret_cast<cf*>(srow)[k] = xa[i];
                    reinterpret_cast<cf*>(srow)[NC - k] = xb[i];
                } else {
                    srow[k] = spectral_row_value<MODE>(xa[i].x, ep);
                    srow[NC - k] = spectral_row_value<MODE>(xa[i].y, ep);
                }
            }
            if (t == 0) {
                if constexpr (MODE == 0) reinterpret_cast<cf*>(srow)[NC / 2] = xm;
                else srow[NC / 2] = spectral_row_value<MODE>(cnorm2(xm), ep);
            }
            wave_lds_fence();
        }
        // ---- the next unit's samples go out now (the FFT's registers are free), before this unit's stores
        __builtin_amdgcn_sched_barrier(0);
        request(nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int nlive = (T - uframe0) < G ? (T - uframe0) : G;
        if constexpr (MEL) {
            const int am = (int)(g0 & 3);
            float* const mstage = reinterpret_cast<float*>(wbase) + MEL_OFF + am;
            lane_mel_contract<S, LPF, SM_FLY>(srow, NC + 1, mlo, mwl, t, mel, mstage + sub * mel.n_mels);
            wave_lds_fence();
            lane_mel_store<(G * LM_MAX_MELS + 255) / 256>(mstage, am, nlive * mel.n_mels, mel.out + g0, lane);
            wave_lds_fence();   // next iteration's first-pass writes must follow these reads
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
            const f4* const s4 = reinterpret_cast<const f4*>(stage + npre);
            f4* const g4 = reinterpret_cast<f4*>(gdst + npre);
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
            wave_lds_fence();   // next iteration's first-pass writes must follow these reads
        }
        unit = nxt;
    }
}

}  // namespace tac
