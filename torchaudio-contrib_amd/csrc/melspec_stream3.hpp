// melspec_stream3.hpp — the fused fft_length-2048 chain (STFT -> |X|^p -> band-sparse mel filterbank -> dB, one launch; the
// benchmark kernel since round 3) with THREE waves per SIMD.  Replaces reference layers.py:307-381 / functional.py:36-38,
// 58-72, 172-184, 291-296 for this size.
//
// melspec_stream_kernel (round 2) runs two 239-register waves per SIMD, each rotating two frames; its counters say that both
// the VALU (64 %) and the LDS (44 %) idle while the SIMD's two waves are stuck at the same time.  This form trades the
// software rotation for a third hardware wave: twelve waves per workgroup (<= 168 VGPRs), ONE frame per wave, the
// |X|^2 row written in place over the frame's exchange area (which is what lets twelve waves fit the LDS:
// 12 x 8.7 KB + 20 KB of weights), no instruction of one frame interleaved with another's.
#pragma once
#include "melspec_stream.hpp"

namespace tac {

constexpr int S3_WAVES = 12;          // what ships
constexpr int S3_WAVES_F32 = 15;      // A/B form (TAC_S3_WAVES=15): as many 128-register waves as the LDS holds next to a 128-band bank

// exchange areas + bank weights + pass-1 twiddles + frame counter + R2C twiddle table + window table [+ mu-law table]
template <int NC, int E>
__host__ __device__ inline size_t stream3_lds_bytes(int wtot, int waves, bool coded) {
    using C = StreamCfg<NC, E>;
    size_t xa = ((size_t)C::F::PADDED * sizeof(cf) + 15) & ~(size_t)15;
    return (size_t)waves * xa + (((size_t)wtot * 4 + 15) & ~(size_t)15) + ST_TW_BYTES + 16 + 64 * (C::F::NPAIR + E) * sizeof(cf) + (coded ? 1024 : 0);
}

#ifndef TAC_S3_PTW_REGS
#define TAC_S3_PTW_REGS 0
#endif
#ifndef TAC_S3_LDS_EXCHANGE
#define TAC_S3_LDS_EXCHANGE 0
#endif
#ifndef TAC_S3_CHUNK
#define TAC_S3_CHUNK 5
#endif
#ifndef TAC_S3_WIN_REGS
#define TAC_S3_WIN_REGS 0
#endif
// FAST1: 0 = any bank the lane layout takes; otherwise the number of steps of slot 1 in the (4, FAST1)-step two-slot layout of
// a 128-band bank (what tac_melbank_pack produces for the standard mel banks): the contraction fully unrolled
// FMT: sample format of the frame load (FMT_*): float32, int16 PCM, mu-law codes as uint8 / int64 (converted in registers)
// WAVES: waves per workgroup (= per CU).  12 is three per SIMD; 15 (four on three of the SIMDs: what the LDS holds next to a
// 128-band bank) caps the registers at 128, which costs a handful of loop-invariant reloads per frame
template <int NC, int E, bool POW2, int FMT, int FAST1, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (WAVES + 3) / 4)
melspec_stream3_kernel(FrameGeom g, Tables tb, StreamArgs m) {
    using C = StreamCfg<NC, E>;
    using F = typename C::F;
    constexpr int NBINS = C::NBINS;
    constexpr int XA_BYTES = (F::PADDED * sizeof(cf) + 15) & ~15;
    static_assert((int)(F::PADDED * sizeof(cf)) >= (int)(C::PROW * 4), "the power row fits the exchange area");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#ifndef TAC_S3_CYCLES
#define TAC_S3_CYCLES 0
#endif
#if TAC_S3_CYCLES
    const unsigned long long wall_entry = wall_clock64();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* const xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);
    float* const prow = reinterpret_cast<float*>(xa);                               // the frame's |X|^2 row, in place
    const long long chunk = (m.total + gridDim.x - 1) / gridDim.x;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < m.total ? begin + chunk : m.total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    const unsigned T = (unsigned)g.n_frames;
    const int t = lane;

#ifndef TAC_S3_ROTATE_DEAL
#define TAC_S3_ROTATE_DEAL 1
#endif
    // the i-th frame dealt is frame (i + nloc / 2) mod nloc of the chunk: a chunk is often a whole row, whose first and last frames
    // touch the padding and take the slower gather path — they are dealt in the middle of the run, not as its tail
    const int deal_shift = TAC_S3_ROTATE_DEAL ? (nloc >> 1) : 0;
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
        const bool ok = g.vec2_ok && start >= 0 && start + F::N <= g.length;
        mode = ok ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + F::N <= g.length ? cs : g.length - F::N;
        const long long off = (long long)row * g.row_stride + cs;                  // in samples
        if constexpr (FMT == FMT_F32) {
            const cf* src = reinterpret_cast<const cf*>(static_cast<const float*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = src[t + q * F::LPF];
        } else if constexpr (FMT == FMT_I16) {                              // a pair of samples = one dword
            const unsigned* src = reinterpret_cast<const unsigned*>(static_cast<const short*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[q].x = __uint_as_float(src[t + q * F::LPF]);
        } else if constexpr (FMT == FMT_MULAW_U8) {                         // a pair of codes = one 16-bit load
            const unsigned short* src = reinterpret_cast<const unsigned short*>(static_cast<const unsigned char*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[q].x = __uint_as_float((unsigned)src[t + q * F::LPF]);
        } else {                                                            // int64 codes: the low dword of each
            const int* src = reinterpret_cast<const int*>(static_cast<const long long*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                v[q].x = __int_as_float(src[4 * (t + q * F::LPF)]);
                v[q].y = __int_as_float(src[4 * (t + q * F::LPF) + 2]);
            }
        }
    };
#ifndef TAC_S3_EARLY_FIRST
#define TAC_S3_EARLY_FIRST 1
#endif
    // ---- tables into LDS.  Every global load of the set-up is issued before the first LDS store (loads of one loop iteration
    //      used to wait for the previous iteration's: a dozen serialized L2 round trips, 4.3 us of a 110 us kernel; now ~one)
    typedef float pf4 __attribute__((ext_vector_type(4)));
    float* const wlds = reinterpret_cast<float*>(smem_raw + (size_t)WAVES * XA_BYTES);
    float* const twlds = wlds + ((m.wtot + 3) & ~3);
    unsigned* const next_frame = reinterpret_cast<unsigned*>(twlds + ST_TW_BYTES / 4);
    cf* const ptwl = reinterpret_cast<cf*>(next_frame + 4);
    cf* const winl = ptwl + 64 * F::NPAIR;
    // (2X -> scale * X once, in the window; int16 PCM samples enter as integers: their 2^-15 goes in as well)
    const float half = 0.5f * g.scale * (FMT == FMT_I16 ? (1.0f / 32768.0f) : 1.0f);
    constexpr int WCH = 4;                                 // 16-byte weight chunks per thread and round
    const int n4 = (m.wtot + 3) >> 2;                      // (the pack is zero-padded to whole 16-byte chunks)
    pf4 wreg[WCH];
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const int c = tid + u * WAVES * 64;
        wreg[u] = reinterpret_cast<const pf4*>(m.wl)[c < n4 ? c : n4 - 1];
    }
    const int js1 = (tid >> 4) & 15, q1 = tid & 15;
    const cf tw1v = tb.w_nc[js1 * q1 * (NC / 256)];
    constexpr int NPT = (64 * F::NPAIR + WAVES * 64 - 1) / (WAVES * 64), NWT = (64 * E + WAVES * 64 - 1) / (WAVES * 64);
    cf ptv[NPT], wnv[NWT];
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int idx = tid + u * WAVES * 64, ic = idx < 64 * F::NPAIR ? idx : 0;
        ptv[u] = tb.w_n[ic / F::NPAIR + (ic % F::NPAIR) * F::LPF];
    }
#pragma unroll
    for (int u = 0; u < NWT; ++u) {
        const int idx = tid + u * WAVES * 64, ic = idx < 64 * E ? idx : 0;
        wnv[u] = window_pair(g, ic / E + (ic % E) * F::LPF);
    }
    cf tw2[3];
    {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw2[q] = all[twiddles_before(NC, E, 2) + q];
    }
    int lo_s[ST_MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < ST_MAX_SLOTS; ++s) lo_s[s] = s < m.nslot ? m.lo[s * 64 + lane] : 0;
    float lutv = 0.0f;
    if (FMT >= FMT_MULAW_U8) lutv = m.lut[tid & 255];
    __builtin_amdgcn_sched_barrier(0);
#if TAC_S3_CYCLES
    const unsigned long long wall_a = wall_clock64();                  // loads issued
    __builtin_amdgcn_s_waitcnt(0x0F70);                                // (debug) vmcnt(0)
    const unsigned long long wall_b = wall_clock64();                  // loads landed
#endif
#if TAC_S3_EARLY_FIRST
    // the wave's first frame is requested BEHIND the table loads (one in-order vmcnt: the tables' wait then leaves these sixteen
    // loads outstanding) and ahead of the LDS stores and the barrier: its HBM latency runs behind the rest of the set-up
    if (nloc > 0) request(w);
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- stores
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const int c = tid + u * WAVES * 64;
        if (c < n4) reinterpret_cast<pf4*>(wlds)[c] = wreg[u];
    }
    for (int c = tid + WCH * WAVES * 64; c < n4; c += WAVES * 64)          // banks with more than 48 KB of weights: the rest, plainly
        reinterpret_cast<pf4*>(wlds)[c] = reinterpret_cast<const pf4*>(m.wl)[c];
    if (tid < 16 * 16) {
        const cf wv = q1 ? tw1v : mkc(1.0f, 0.0f);
        twlds[js1 * ST_TW_STRIDE + 2 * (q1 ? q1 - 1 : 15)] = wv.x;
        twlds[js1 * ST_TW_STRIDE + 2 * (q1 ? q1 - 1 : 15) + 1] = wv.y;
    }
    if (tid == 0) *next_frame = WAVES;
    // the eight R2C twiddles of a lane (re-read every frame: the register budget does not hold them) and the window pairs of its
    // sixteen first-pass elements (scale folded in), as [read u][lane] 16-byte pairs: every ds_read_b128 of the wave is one
    // contiguous kilobyte
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int idx = tid + u * WAVES * 64;
        const int tt = idx / F::NPAIR, p = idx - tt * F::NPAIR;
        if (idx < 64 * F::NPAIR) ptwl[((p >> 1) * 64 + tt) * 2 + (p & 1)] = ptv[u];
    }
#pragma unroll
    for (int u = 0; u < NWT; ++u) {
        const int idx = tid + u * WAVES * 64;
        const int tt = idx / E, q = idx - tt * E;
        if (idx < 64 * E) winl[((q >> 1) * 64 + tt) * 2 + (q & 1)] = cscale(wnv[u], half);
    }
#if TAC_S3_PTW_REGS
    cf ptw_regs[F::NPAIR];
#pragma unroll
    for (int p = 0; p < F::NPAIR; ++p) ptw_regs[p] = tb.w_n[t + p * F::LPF];
#endif
    float* const lutlds = reinterpret_cast<float*>(winl + 64 * E);                 // mu-law decode table (coded inputs)
    if (FMT >= FMT_MULAW_U8 && tid < 256) lutlds[tid] = lutv;
#if TAC_S3_WIN_REGS
    cf win_regs[E];
    load_window_regs<F>(win_regs, g, t);
#pragma unroll
    for (int e = 0; e < E; ++e) win_regs[e] = cscale(win_regs[e], half);
#endif
#if TAC_S3_CYCLES
    const unsigned long long wall_c = wall_clock64();                  // stores issued
#endif
    __syncthreads();
    if (nloc <= 0) return;

    typedef float f4 __attribute__((ext_vector_type(4)));
    const bool fast_db = m.amin >= 1.1754944e-38f;
    const float ten_log10_ref = 10.0f * m.log10_ref;
    // the requested registers as float sample pairs (still unwindowed): PCM integers / decoded codes
    auto decode = [&]() {
        if constexpr (FMT == FMT_I16) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int bits = __float_as_int(v[q].x);
                v[q] = mkc((float)(short)(bits & 0xffff), (float)(bits >> 16));
            }
        } else if constexpr (FMT == FMT_MULAW_U8) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const unsigned bits = __float_as_uint(v[q].x);
                v[q] = mkc(lutlds[bits & 0xffu], lutlds[(bits >> 8) & 0xffu]);
            }
        } else if constexpr (FMT == FMT_MULAW_I64) {
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = mkc(lutlds[__float_as_uint(v[q].x) & 0xffu], lutlds[__float_as_uint(v[q].y) & 0xffu]);
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
    auto fma4 = [](f4 wv, f4 pv, cf& a0, cf& a1) {
        a0 = __builtin_elementwise_fma(mkc(wv.x, wv.y), mkc(pv.x, pv.y), a0);
        a1 = __builtin_elementwise_fma(mkc(wv.z, wv.w), mkc(pv.z, pv.w), a1);
    };
    int i = w;
#if !TAC_S3_EARLY_FIRST
    request(i);
#endif
#ifndef TAC_S3_CYCLES
#define TAC_S3_CYCLES 0
#endif
#if TAC_S3_CYCLES
    const unsigned long long cyc0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
#endif
    while (i < nloc) {
        // the next frame of this wave (the counter's answer travels with the stage's other LDS traffic)
        unsigned ask = 0;
        if (lane == 0) ask = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- s0: window, pass 0, exchange
        if (mode == 1) {
            decode();
        } else {                                            // frames touching the padding gather their (decoded) samples first
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v, g, nullptr, xa, row, fr, tz, Fetch{m.samples, lutlds});
        }
        {
#if TAC_S3_WIN_REGS
            Dft<16>::run_windowed(v, win_regs);
#else
            cf win[E];
            const f4* wl = reinterpret_cast<const f4*>(winl) + t;
#pragma unroll
            for (int u = 0; u < E / 2; ++u) {
                const f4 x = wl[u * 64];
                win[2 * u] = mkc(x.x, x.y);
                win[2 * u + 1] = mkc(x.z, x.w);
            }
            Dft<16>::run_windowed(v, win);
#endif
        }
        wave_lds_fence();
        cf tw1[16];
        {
            const f4* tl = reinterpret_cast<const f4*>(twlds + (t & 15) * ST_TW_STRIDE);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4 x = tl[u];
                tw1[2 * u] = mkc(x.x, x.y);
                tw1[2 * u + 1] = mkc(x.z, x.w);
            }
        }
        F::template pass_write<0, true>(v, xa, t, t);
        wave_lds_fence();
        F::template pass_readback<1>(v, xa, t);
        // ---- s12
        F::template pass_twiddle<1, true>(v, tw1);
        F::template pass_butterflies<1>(v);
#if TAC_S3_LDS_EXCHANGE
        wave_lds_fence();                                   // (A/B) the pass 1 -> 2 exchange through the LDS area instead of permlane swaps
        F::template pass_write<1, true>(v, xa, t, t);
        wave_lds_fence();
        F::template pass_readback<2>(v, xa, t);
#else
        F::exchange_1_2_in_registers(v);
#endif
        F::template pass_twiddle<2, true>(v, tw2);
        F::template pass_butterflies<2>(v);
        wave_lds_fence();
        F::template pass_write<2, true>(v, xa, t, t);
        wave_lds_fence();
        cf zm[F::NPAIR], zmid;
        {
            const cf* const pb = xa + lds_pad(NC - t);
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const cf z = pb[-lds_pad_c(p * F::LPF)];
                zm[p] = (p == 0 && t == 0) ? v[F::reg_of_spectrum(0)] : z;
            }
            zmid = xa[lds_pad(NC / 2)];
        }
        // ---- s3: R2C split -> |X|^p; the row overwrites the exchange area once every lane holds its partners
        cf pw[F::NPAIR];
#if TAC_S3_PTW_REGS
        const cf (&ptw)[F::NPAIR] = ptw_regs;
#else
        cf ptw[F::NPAIR];
        {
            const f4* pl = reinterpret_cast<const f4*>(ptwl) + t;
#pragma unroll
            for (int u = 0; u < F::NPAIR / 2; ++u) {
                const f4 x = pl[u * 64];
                ptw[2 * u] = mkc(x.x, x.y);
                ptw[2 * u + 1] = mkc(x.z, x.w);
            }
        }
#endif
#pragma unroll
        for (int p = 0; p < F::NPAIR; p += 2)
            r2c_power_pair_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], v[F::reg_of_spectrum(p + 1)], zm[p + 1], ptw[p + 1], pw[p], pw[p + 1]);
        const float pmid = 4.0f * cnorm2(zmid);
        wave_lds_fence();                                                   // all partner reads are in registers
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            const int kk = t + p * F::LPF;
            prow[kk] = POW2 ? pw[p].x : __builtin_amdgcn_sqrtf(pw[p].x);
            prow[NC - kk] = POW2 ? pw[p].y : __builtin_amdgcn_sqrtf(pw[p].y);
        }
        if (t == 0) prow[NC / 2] = POW2 ? pmid : __builtin_amdgcn_sqrtf(pmid);
        if (t < C::PROW - NBINS) prow[NBINS + t] = 0.0f;                    // slack taps carry zero weights: keep them finite
        wave_lds_fence();
        // ---- the next frame's samples go out now (v is dead), they land during the contraction
        const int cur = i;
        i = (int)__builtin_amdgcn_readfirstlane(ask);
        request(i);
        // ---- s4: filterbank contraction, dB, row store
        if constexpr (FAST1 > 0) {
            const int ci = cur < nloc ? cur : nloc - 1;
            const f4* wp = reinterpret_cast<const f4*>(wlds) + lane;
            const f4* p0 = reinterpret_cast<const f4*>(prow + lo_s[0]);
            const f4* p1 = reinterpret_cast<const f4*>(prow + lo_s[1]);
            cf a0 = mkc(0.f, 0.f), a1 = mkc(0.f, 0.f), b0 = mkc(0.f, 0.f), b1 = mkc(0.f, 0.f);
            if constexpr ((WAVES + 3) / 4 >= 4) {
            // four waves per SIMD leave 128 registers: the taps arrive in chunks of TAC_S3_CHUNK steps, the other waves cover the waits
            {
                f4 w0[ST_FAST_STEPS0], q0[ST_FAST_STEPS0];
#pragma unroll
                for (int u = 0; u < ST_FAST_STEPS0; ++u) {
                    w0[u] = wp[u * 64];
                    q0[u] = p0[u];
                }
#pragma unroll
                for (int u = 0; u < ST_FAST_STEPS0; ++u) fma4(w0[u], q0[u], a0, a1);
            }
#pragma unroll
            for (int c = 0; c < FAST1; c += TAC_S3_CHUNK) {
                constexpr int CH = TAC_S3_CHUNK;
                f4 wc[CH], qc[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (c + u < FAST1) {
                        wc[u] = wp[(ST_FAST_STEPS0 + c + u) * 64];
                        qc[u] = p1[c + u];
                    }
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (c + u < FAST1) fma4(wc[u], qc[u], b0, b1);
            }
            } else {
            constexpr int B1 = (FAST1 + 1) / 2, B2 = FAST1 - B1;              // slot 1 in two batches
            f4 w0[ST_FAST_STEPS0], q0[ST_FAST_STEPS0], wa[B1], qa[B1];
#pragma unroll
            for (int u = 0; u < ST_FAST_STEPS0; ++u) {
                w0[u] = wp[u * 64];
                q0[u] = p0[u];
            }
#pragma unroll
            for (int u = 0; u < B1; ++u) {
                wa[u] = wp[(ST_FAST_STEPS0 + u) * 64];
                qa[u] = p1[u];
            }
#pragma unroll
            for (int u = 0; u < ST_FAST_STEPS0; ++u) fma4(w0[u], q0[u], a0, a1);
            f4 wb[B2], qb[B2];
#pragma unroll
            for (int u = 0; u < B2; ++u) {
                wb[u] = wp[(ST_FAST_STEPS0 + B1 + u) * 64];
                qb[u] = p1[B1 + u];
            }
#pragma unroll
            for (int u = 0; u < B1; ++u) fma4(wa[u], qa[u], b0, b1);
#pragma unroll
            for (int u = 0; u < B2; ++u) fma4(wb[u], qb[u], b0, b1);
            }
            float v0 = (a0.x + a0.y) + (a1.x + a1.y), v1 = (b0.x + b0.y) + (b1.x + b1.y);
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
            float* orow = m.out + (begin + place(ci)) * (long long)m.n_mels + lane;
#pragma unroll
            for (int s = 0; s < ST_MAX_SLOTS; ++s) {
                if (s < m.nslot) {
                    const f4* pp = reinterpret_cast<const f4*>(prow + lo_s[s]);
                    const int n = m.steps[s];
                    cf acc0 = mkc(0.f, 0.f), acc1 = mkc(0.f, 0.f);
                    int j = 0;
#pragma unroll 1
                    for (; j + 8 <= n; j += 8, wp += 512) {
                        f4 wv[8], pv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            wv[u] = wp[u * 64];
                            pv[u] = pp[j + u];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) fma4(wv[u], pv[u], acc0, acc1);
                    }
                    if (j + 4 <= n) {
                        f4 wv[4], pv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            wv[u] = wp[u * 64];
                            pv[u] = pp[j + u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) fma4(wv[u], pv[u], acc0, acc1);
                        wp += 256;
                        j += 4;
                    }
#pragma unroll 1
                    for (; j < n; ++j, wp += 64) fma4(wp[0], pp[j], acc0, acc1);
                    float val = (acc0.x + acc0.y) + (acc1.x + acc1.y);
                    if (m.db) val = fast_db ? amp_to_db_fast(val, m.amin, ten_log10_ref) : amp_to_db(val, m.amin, m.log10_ref);
                    if (s * 64 + lane < m.n_mels) orow[s * 64] = val;
                }
            }
        }
        wave_lds_fence();                                                   // the row is consumed: the area takes the next frame
    }
#if TAC_S3_CYCLES
    // debug build (tools/stream3_cycles.py): shader cycles and 100 MHz ticks of every wave's frame loop, over the first outputs
    __syncthreads();
    if (lane == 0) {
        m.out[((long long)blockIdx.x * WAVES + w) * 2] = (float)(__builtin_readcyclecounter() - cyc0);
        m.out[((long long)blockIdx.x * WAVES + w) * 2 + 1] = (float)(wall_clock64() - wall0);
        if (w == 0) m.out[(long long)gridDim.x * WAVES * 2 + blockIdx.x] = (float)(wall0 - wall_entry);   // set-up: entry -> loop
        if (w == 0 && blockIdx.x == 7) {
            m.out[(long long)gridDim.x * WAVES * 2 + gridDim.x + 0] = (float)(wall_a - wall_entry);
            m.out[(long long)gridDim.x * WAVES * 2 + gridDim.x + 1] = (float)(wall_b - wall_a);
            m.out[(long long)gridDim.x * WAVES * 2 + gridDim.x + 2] = (float)(wall_c - wall_b);
            m.out[(long long)gridDim.x * WAVES * 2 + gridDim.x + 3] = (float)(wall0 - wall_c);
        }
    }
#endif
}

}  // namespace tac
