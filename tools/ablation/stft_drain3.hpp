// stft_drain3.hpp — the fft_length-2048 STFT / spectrogram rows with WAVE-SPECIALISED STORES (round 5).
//
// stft_stream3_kernel lets every wave transform a frame AND stream its row out: gfx950 counts loads and stores in one in-order
// vmcnt, a CU's twelve / sixteen waves block in turn on store issue, and the kernel runs at "transform + stores" instead of
// max(transform, stores).  Here a workgroup is TW transform waves + DW drain waves:
//
//   * a transform wave never touches global memory for writing.  It puts the finished row of frame i into slot i mod NS of a
//     RING in LDS and raises the slot's state; the ring is TIGHTLY PACKED — NS * LENF floats, rows back to back exactly as they
//     lie in the frame-major output (out[rows][T][F] is one contiguous stream per workgroup: frame begin + i follows frame
//     begin + i - 1) — and starts at the 16-byte phase of the workgroup's first output element, so that every 16-byte piece of
//     the OUTPUT STREAM is a 16-byte aligned piece of the ring, row boundaries included;
//   * a drain wave stores SPANS of that stream: span j = [B_j, B_j+1) with B_j = the row start of frame j rounded DOWN to a
//     1 KB boundary of the output (the workgroup's first / last span start / end at the chunk's own bounds).  An interior span
//     is 8 or 9 (complex rows; 4 or 5 for real rows) wave-stores of 16 bytes per lane, every one a whole, 1 KB-aligned
//     kilobyte: each cache line of the output is written exactly once, by one instruction (the per-row form needs 11 store
//     instructions per 8200-byte row, two of them 4-byte head / tail stores, and nearly every row shares its first and last
//     line with its neighbours).  Span j holds the tail of row j - 1 and all but the tail of row j.
//
// Slot protocol (one LDS word per slot, three events per lap): the producer of row i (slot s = i mod NS, lap L = i / NS) waits
// for state[s] == 3 L, writes the row, waits for its LDS writes (lgkmcnt) and sets 3 L + 1; spans i and i + 1 each wait for
// >= 3 L + 1, read their pieces into registers, wait for the reads and add 1.  The smallest unproduced row only ever waits for
// spans that depend on rows below it (NS >= 2), so the protocol cannot deadlock; every wait is bounded anyway (a protocol bug
// would give wrong rows, which the tests catch, instead of hanging the GPU).
// Replaces torch.stft (reference functional.py:99-107) [+ complex_norm (functional.py:126-128)] [+ amplitude_to_db].
#pragma once
#include "stft_stream3.hpp"

#ifndef TAC_S3_DRAIN
#define TAC_S3_DRAIN 0            // 1: launch_pipe3 (stft_kernels.hip) takes this form
#endif
#ifndef TAC_S3_DRAIN_TW_C
#define TAC_S3_DRAIN_TW_C 12      // complex rows: transform + drain waves
#define TAC_S3_DRAIN_DW_C 4
#endif
#ifndef TAC_S3_DRAIN_PRIO
#define TAC_S3_DRAIN_PRIO 0       // A/B: s_setprio of the drain waves
#endif
#ifndef TAC_S3_DRAIN_ABL
#define TAC_S3_DRAIN_ABL 0        // timing-only ablations (WRONG RESULTS): 1 the drain waves read their spans but store nothing,
#endif                            // 2 the transform waves skip window + transform (sample loads, row writes and the stores remain), 3 the transform waves never wait for a slot
#ifndef TAC_S3_DRAIN_SLEEP
#define TAC_S3_DRAIN_SLEEP 1      // s_sleep argument between two polls of a slot state
#endif
#ifndef TAC_S3_DRAIN_INPLACE
#define TAC_S3_DRAIN_INPLACE 1    // 1: stft_drain3i_kernel (rows in place, static dealing, tail buffers), 0: stft_drain3_kernel (LDS ring)
#endif
#ifndef TAC_S3_DRAIN_STAMPS
#define TAC_S3_DRAIN_STAMPS 0     // debug build of tools/r05/drain_stamps.py (WRONG RESULTS): per-wave cycle sums overwrite the head of out[]
#endif
#if TAC_S3_DRAIN_STAMPS
#define D3_STAMP(i)                                                          \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();        \
        stamp_acc[i] += now_ - stamp_last;                                   \
        stamp_last = now_;                                                   \
        __builtin_amdgcn_sched_barrier(0);                                   \
    } while (0)
#define D3_STAMP_INIT unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long stamp_last = __builtin_readcyclecounter(); const unsigned long long stamp_first = stamp_last
#define D3_STAMP_FLUSH                                                                                                    \
    do {                                                                                                                  \
        stamp_acc[7] = __builtin_readcyclecounter() - stamp_first;                                                        \
        if (t < 8) ep.out[((long long)blockIdx.x * WAVES + w) * 8 + t] = (float)stamp_acc[t];                              \
    } while (0)
#else
#define D3_STAMP(i) do { } while (0)
#define D3_STAMP_INIT do { } while (0)
#define D3_STAMP_FLUSH do { } while (0)
#endif
#ifndef TAC_S3_DRAIN_PIPE
#define TAC_S3_DRAIN_PIPE 1       // drain waves ask for the next span's rows and read them behind the current span's stores
#endif
#ifndef TAC_S3_DRAIN_LATEPUB
#define TAC_S3_DRAIN_LATEPUB 0    // A/B (ring form): a row is published behind the NEXT frame's first butterfly instead of behind a wait for its own LDS writes
#endif
#ifndef TAC_S3_DRAIN_TW_R
#define TAC_S3_DRAIN_TW_R 12      // real rows
#define TAC_S3_DRAIN_DW_R 4
#endif

namespace tac {

template <int NC, int E, int MODE, int TW, int DW>
struct Drain3Cfg {
    using F = WaveFft<NC, E>;
    static constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);              // floats per output row
    static constexpr int XA = s3_xa_bytes<F>();
    static constexpr int TABLES = ST_TW_BYTES + 64 + 64 * (F::NPAIR + E) * (int)sizeof(cf);
    static constexpr int CTRL = 128;                                         // slot states (<= 32 slots)
    static constexpr int LDS_MAX = 160 * 1024;
    static constexpr int MULT = MODE == 0 ? 2 : 4;                           // NS * LENF * 4 must be a multiple of 16 bytes
    static constexpr int FREE = LDS_MAX - TW * XA - TABLES - CTRL - 32;
    static constexpr int NS_FIT = (FREE / (LENF * 4)) / MULT * MULT;
    static constexpr int NS = NS_FIT > 32 ? 32 : NS_FIT;
    static constexpr int RF = NS * LENF;                                     // floats in the ring
    static constexpr int BYTES = TW * XA + TABLES + CTRL + (RF + 8) * 4;     // + phase (<= 3 floats) + mirror (4 floats)
    static constexpr int NSTMAX = (LENF + 255) / 256 + 1;                    // wave-stores of a span, at most
    static_assert(NS >= 2 && (RF % 4) == 0 && BYTES <= LDS_MAX, "ring");
};

// state >= want (wrap-safe), polled by the whole wave (one broadcast read); bounded
__device__ __forceinline__ void drain3_wait(unsigned addr, unsigned want) {
    for (int guard = 0; guard < (1 << 17); ++guard) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        if ((int)(v - want) >= 0) break;
        if (TAC_S3_DRAIN_SLEEP) __builtin_amdgcn_s_sleep(TAC_S3_DRAIN_SLEEP);
    }
}

// two states in one round trip
__device__ __forceinline__ void drain3_wait2(unsigned addr_a, unsigned want_a, unsigned addr_b, unsigned want_b) {
    for (int guard = 0; guard < (1 << 17); ++guard) {
        unsigned va, vb;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(va), "=&v"(vb) : "v"(addr_a), "v"(addr_b) : "memory");
        va = (unsigned)__builtin_amdgcn_readfirstlane((int)va);
        vb = (unsigned)__builtin_amdgcn_readfirstlane((int)vb);
        if ((int)(va - want_a) >= 0 && (int)(vb - want_b) >= 0) break;
        if (TAC_S3_DRAIN_SLEEP) __builtin_amdgcn_s_sleep(TAC_S3_DRAIN_SLEEP);
    }
}

// MODE as in stft_stream3_kernel; TW + DW = 12 (168 registers: the transform waves request the next frame a whole frame ahead)
// or 16 (128 registers, late request)
template <int NC, int E, int MODE, int TW, int DW>
__global__ void __launch_bounds__((TW + DW) * 64, (TW + DW) / 4)
stft_drain3_kernel(FrameGeom g, Tables tb, StftEpilogue ep, Stream3Launch lp) {
    using F = WaveFft<NC, E>;
    using D = Drain3Cfg<NC, E, MODE, TW, DW>;
    static_assert(F::G == 1 && E == 16 && radix_at(NC, 0) == 16, "fft_length 2048");
    
    constexpr int WAVES = TW + DW;
    constexpr int XA_BYTES = D::XA;
    constexpr int LENF = D::LENF;
    constexpr int NS = D::NS;
    constexpr int RF = D::RF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const twlds = reinterpret_cast<float*>(smem_raw + (size_t)TW * XA_BYTES);
    unsigned* const next_frame = reinterpret_cast<unsigned*>(twlds + ST_TW_BYTES / 4);
    cf* const ptwl = reinterpret_cast<cf*>(next_frame + 16);
    cf* const winl = ptwl + 64 * F::NPAIR;
    unsigned* const state = reinterpret_cast<unsigned*>(winl + 64 * E);
    float* const ringmem = reinterpret_cast<float*>(state + D::CTRL / 4);
    const float half = 0.5f * g.scale;
    S3Setup<F, WAVES * 64> setup;
    setup.issue(g, tb, tid);

    const long long total = g.rows * g.n_frames;
    const long long chunk = lp.chunk;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < total ? begin + chunk : total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    const unsigned T = (unsigned)g.n_frames;
    // absolute float index (address / 4) of the workgroup's first output element; the ring starts at its 16-byte phase
    const long long X0 = (long long)(reinterpret_cast<unsigned long long>(ep.out) >> 2) + begin * LENF;
    float* const ring = ringmem + (int)(X0 & 3);
    const unsigned state_addr = lds_offset_of(state);
    typedef float f4 __attribute__((ext_vector_type(4)));

    if (w >= TW) {
        // =================================================== drain waves
        if (tid - TW * 64 < D::CTRL / 4) state[tid - TW * 64] = 0u;
        setup.store(twlds, ptwl, winl, half, tid);
        __syncthreads();
        const int d = w - TW;
        if (TAC_S3_DRAIN_PRIO) __builtin_amdgcn_s_setprio(TAC_S3_DRAIN_PRIO);
        D3_STAMP_INIT;
        // a span's geometry (all wave-uniform)
        struct Span {
            long long lo;
            int j, npre, nch, ntail, S, NI, rlo, s1, s0;
            unsigned want1, want0;
        };
        auto make = [&](int j) {
            Span sp;
            sp.j = j;
            sp.s1 = j % NS;
            sp.s0 = j > 0 ? (j - 1) % NS : sp.s1;
            sp.want1 = 3u * (unsigned)(j / NS) + 1u;
            sp.want0 = j > 0 ? 3u * (unsigned)((j - 1) / NS) + 1u : sp.want1;
            const long long Xj = X0 + (long long)j * LENF;
            sp.lo = j == 0 ? X0 : (Xj & ~255LL);
            const long long hi = j == nloc - 1 ? Xj + LENF : ((Xj + LENF) & ~255LL);
            int rlo = sp.s1 * LENF - (int)(Xj - sp.lo);           // ring offset of the span's first float
            sp.rlo = rlo < 0 ? rlo + RF : rlo;
            const long long lo16 = (sp.lo + 3) & ~3LL;
            sp.npre = (int)(lo16 - sp.lo);
            const int nbody = (int)(hi - lo16);
            sp.nch = nbody >> 2;
            sp.ntail = nbody & 3;
            sp.S = (int)(lo16 >> 2) & 63;                          // chunks between the previous 1 KB boundary and the span's first chunk
            sp.NI = (sp.nch + sp.S + 63) >> 6;
            return sp;
        };
        auto wait_rows = [&](const Span& sp) {
            drain3_wait2(state_addr + 4u * (unsigned)sp.s1, sp.want1, state_addr + 4u * (unsigned)sp.s0, sp.want0);
        };
        auto wrap = [](int r) { return r >= RF ? r - RF : r; };
        // the span's LDS reads (issued, not waited for)
        auto read_span = [&](const Span& sp, f4 (&bb)[D::NSTMAX], float& hv, float& tv) {
            hv = 0.0f;
            tv = 0.0f;
            if (sp.npre) hv = ring[wrap(sp.rlo + (t < sp.npre ? t : 0))];
            if (sp.ntail) tv = ring[wrap(sp.rlo + sp.npre + 4 * sp.nch + (t < sp.ntail ? t : 0))];
            const int rb = sp.rlo + sp.npre;
#pragma unroll
            for (int u = 0; u < D::NSTMAX; ++u) {
                if (u < sp.NI) {
                    int c = t + 64 * u - sp.S;
                    c = c < 0 ? 0 : (c < sp.nch ? c : sp.nch - 1);
                    bb[u] = *reinterpret_cast<const f4*>(ring + wrap(rb + 4 * c));
                }
            }
        };
        // ... landed in registers: both slots get their event
        auto release = [&](const Span& sp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t == 0) {
                __hip_atomic_fetch_add(state + sp.s1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (sp.j > 0) __hip_atomic_fetch_add(state + sp.s0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        // wave-stores [u0, u1) of the span (+ its head floats with the first, its tail floats with the last)
        auto store_span = [&](const Span& sp, const f4 (&bb)[D::NSTMAX], float hv, float tv, int u0, int u1) {
#if TAC_S3_DRAIN_ABL == 1
            if (hv + tv + bb[0].x != 12345.678f) return;
#endif
            float* const gp = reinterpret_cast<float*>(static_cast<unsigned long long>(sp.lo) << 2);
            if (u0 == 0 && sp.npre && t < sp.npre) gp[t] = hv;
            f4* const g4 = reinterpret_cast<f4*>(gp + sp.npre);
#pragma unroll
            for (int u = 0; u < D::NSTMAX; ++u) {
                if (u >= u0 && u < u1 && u < sp.NI) {
                    const int c = t + 64 * u - sp.S;
                    if (c >= 0 && c < sp.nch) {
                        if (lp.plain_stores) g4[c] = bb[u];
                        else __builtin_nontemporal_store(bb[u], &g4[c]);
                    }
                }
            }
            if (u1 == D::NSTMAX && sp.ntail && t < sp.ntail) gp[sp.npre + 4 * sp.nch + t] = tv;
        };
#if TAC_S3_DRAIN_PIPE
        // software pipeline: the NEXT span's row states are asked for before the current span's first stores, its reads are in
        // flight behind the rest of them — the LDS round trips (~1000 cycles per span under load) hide behind store issue
        if (d < nloc) {
            f4 B0[D::NSTMAX], B1[D::NSTMAX];
            float h0, t0, h1, t1;
            Span cur = make(d);
            wait_rows(cur);
            D3_STAMP(0);
            read_span(cur, B0, h0, t0);
            release(cur);
            D3_STAMP(1);
            constexpr int USPLIT = D::NSTMAX / 2;
            auto step = [&](const f4 (&bc)[D::NSTMAX], float hc, float tc, f4 (&bn)[D::NSTMAX], float& hn, float& tn) {
                const int jn = cur.j + DW;
                const bool more = jn < nloc;
                const Span nx = more ? make(jn) : cur;
                unsigned va = 0, vb = 0;
                bool got = false;
                if (more) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(va), "=&v"(vb) : "v"(state_addr + 4u * (unsigned)nx.s1), "v"(state_addr + 4u * (unsigned)nx.s0) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                store_span(cur, bc, hc, tc, 0, USPLIT);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va), "+v"(vb) :: "memory");
                    const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane((int)va), sb = (unsigned)__builtin_amdgcn_readfirstlane((int)vb);
                    got = (int)(sa - nx.want1) >= 0 && (int)(sb - nx.want0) >= 0;
                    if (got) read_span(nx, bn, hn, tn);
                }
                __builtin_amdgcn_sched_barrier(0);
                store_span(cur, bc, hc, tc, USPLIT, D::NSTMAX);
                __builtin_amdgcn_sched_barrier(0);
                D3_STAMP(3);
                if (more) {
                    if (!got) {
                        wait_rows(nx);
                        D3_STAMP(0);
                        read_span(nx, bn, hn, tn);
                    }
                    release(nx);
                    D3_STAMP(1);
                }
                cur = nx;
                return more;
            };
            for (;;) {
                if (!step(B0, h0, t0, B1, h1, t1)) break;
                if (!step(B1, h1, t1, B0, h0, t0)) break;
            }
        }
#else
        for (int j = d; j < nloc; j += DW) {
            D3_STAMP(3);                                      // store issue (+ loop overhead)
            const Span sp = make(j);
            wait_rows(sp);
            D3_STAMP(0);                                      // waiting for the rows
            f4 bb[D::NSTMAX];
            float hv, tv;
            read_span(sp, bb, hv, tv);
            release(sp);                                      // the slots are free again once the reads have landed in registers
            D3_STAMP(1);                                      // span reads
            __builtin_amdgcn_sched_barrier(0);
            store_span(sp, bb, hv, tv, 0, D::NSTMAX);
        }
#endif
        D3_STAMP(3);
        D3_STAMP_FLUSH;
        return;
    }

    // ======================================================= transform waves
    cf* const xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);
    cf tw2[3];
    {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw2[q] = all[twiddles_before(NC, E, 2) + q];
    }
    cf v[E];
    int mode = 0, row = 0;
    long long fr = 0;
    constexpr bool EARLY = WAVES == 12;
    cf nx[EARLY ? E : 1];
    int nmode = 0, nrow = 0;
    long long nfr = 0;
    auto request_into = [&](int i, cf* dst, int& mode_o, int& row_o, long long& fr_o) {
        i = i < nloc ? i : nloc - 1;
        const unsigned gf = (unsigned)(begin + i);
        const unsigned r = gf / T;
        row_o = (int)r;
        fr_o = (long long)(gf - r * T);
        const long long start = fr_o * (long long)g.hop - g.center_pad;
        const bool ok = g.vec2_ok && start >= 0 && start + F::N <= g.length;
        mode_o = ok ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + F::N <= g.length ? cs : g.length - F::N;
        const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)row_o * g.row_stride + cs);
#pragma unroll
        for (int q = 0; q < E; ++q) dst[q] = src[t + q * F::LPF];
    };
    auto request = [&](int i) {
        if constexpr (EARLY) request_into(i, nx, nmode, nrow, nfr);
        else request_into(i, v, mode, row, fr);
    };
    if (nloc > 0) request(w);
    __builtin_amdgcn_sched_barrier(0);
    if (tid == 0) *next_frame = TW;
    setup.store(twlds, ptwl, winl, half, tid);
    __syncthreads();
    if (nloc <= 0) return;

    S3Swz swz;
    swz.init(xa, t);
    int i = w;
    unsigned pend_addr = 0, pend_val = 0;                     // TAC_S3_DRAIN_LATEPUB: the previous row's publication, still owed
    D3_STAMP_INIT;
    while (i < nloc) {
#if TAC_S3_DRAIN_STAMPS
        D3_STAMP(3);                                          // (row writes, publish, loop)
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
        D3_STAMP(0);                                          // waiting for the samples
#endif
        unsigned ask = 0;
        int i_next = 0;
        if (t == 0) ask = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (EARLY) {
            mode = nmode;
            row = nrow;
            fr = nfr;
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = nx[q];
        }
        // ---- s0: window, pass 0, exchange
        if (mode != 1) {
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v, g, nullptr, xa, row, fr, tz, FetchF32{g.wave});
        }
#if TAC_S3_DRAIN_ABL == 2
        cf zm[F::NPAIR], zmid = v[0], ptw[F::NPAIR];
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            zm[p] = v[F::NPAIR + p];
            ptw[p] = mkc(1.0f, 0.0f);
        }
        if constexpr (EARLY) {
            i_next = (int)__builtin_amdgcn_readfirstlane(ask);
            __builtin_amdgcn_sched_barrier(0);
            request(i_next);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        {
            cf win[E];
            const f4* wl = reinterpret_cast<const f4*>(winl) + t;
#pragma unroll
            for (int u = 0; u < E / 2; ++u) {
                const f4 x = wl[u * 64];
                win[2 * u] = mkc(x.x, x.y);
                win[2 * u + 1] = mkc(x.z, x.w);
            }
            Dft<16>::run_windowed(v, win);
        }
        D3_STAMP(4);                                          // window, first butterfly
        if constexpr (EARLY) {
            i_next = (int)__builtin_amdgcn_readfirstlane(ask);
            __builtin_amdgcn_sched_barrier(0);
            request(i_next);
            __builtin_amdgcn_sched_barrier(0);
        }
        D3_STAMP(6);                                          // issuing the next frame's sample request (EARLY form)
        if (TAC_S3_DRAIN_LATEPUB && pend_val) {               // (LDS operations of a wave complete in order: the window reads are back, so the row is written)
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"(pend_addr), "v"(pend_val) : "memory");
            pend_val = 0;
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
        s3_write_pass0_swz(v, swz);
        wave_lds_fence();
        s3_readback_pass1_swz(v, swz);
        // ---- s12
        F::template pass_twiddle<1, true>(v, tw1);
        F::template pass_butterflies<1>(v);
        F::exchange_1_2_in_registers(v);
        F::template pass_twiddle<2, true>(v, tw2);
        F::template pass_butterflies<2>(v);
        cf zm[F::NPAIR], zmid;
        s3_r2c_partners<F>(v, xa, zm, zmid, t);
        // ---- s3: R2C split; the row goes to its ring slot
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
        D3_STAMP(5);                                          // exchange, passes 1 and 2, partners
        const int slot = i % NS, lap = i / NS;
        float* const stage = ring + slot * LENF;
        const unsigned st_addr = state_addr + 4u * (unsigned)slot;
        if constexpr (MODE == 0) {
            cf xlo[F::NPAIR], xhi[F::NPAIR], xm, unused;
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) F::r2c_split_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], xlo[p], xhi[p]);
            F::r2c_split_x2(zmid, zmid, mkc(0.0f, -1.0f), xm, unused);
            D3_STAMP(2);                                      // R2C split
            if (TAC_S3_DRAIN_ABL != 3) drain3_wait(st_addr, 3u * (unsigned)lap);
            D3_STAMP(1);                                      // waiting for the slot
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int kk = t + p * F::LPF;
                reinterpret_cast<cf*>(stage)[kk] = xlo[p];
                reinterpret_cast<cf*>(stage)[NC - kk] = xhi[p];
            }
            if (t == 0) reinterpret_cast<cf*>(stage)[NC / 2] = xm;
            if (slot == 0 && t < 2) reinterpret_cast<cf*>(ring + RF)[t] = xlo[0];      // mirror of the ring's first 16 bytes
        } else {
            cf pw[F::NPAIR];
#pragma unroll
            for (int p = 0; p < F::NPAIR; p += 2)
                r2c_power_pair_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], v[F::reg_of_spectrum(p + 1)], zm[p + 1], ptw[p + 1], pw[p], pw[p + 1]);
            const float pmid = 4.0f * cnorm2(zmid);
            D3_STAMP(2);                                      // R2C split
            if (TAC_S3_DRAIN_ABL != 3) drain3_wait(st_addr, 3u * (unsigned)lap);
            D3_STAMP(1);                                      // waiting for the slot
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int kk = t + p * F::LPF;
                const float lov = spectral_row_value<MODE>(pw[p].x, ep);
                stage[kk] = lov;
                stage[NC - kk] = spectral_row_value<MODE>(pw[p].y, ep);
                if (p == 0 && slot == 0 && t < 4) ring[RF + t] = lov;
            }
            if (t == 0) stage[NC / 2] = spectral_row_value<MODE>(pmid, ep);
        }
        // ---- the next frame's samples (late form), then the slot is published once the row's LDS writes have completed
        if constexpr (EARLY) {
            i = i_next;
        } else {
            i = (int)__builtin_amdgcn_readfirstlane(ask);
            __builtin_amdgcn_sched_barrier(0);
            request(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const unsigned val = 3u * (unsigned)lap + 1u;
            if (TAC_S3_DRAIN_LATEPUB) {
                pend_addr = st_addr;
                pend_val = val;
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"(st_addr), "v"(val) : "memory");
            }
        }
        wave_lds_fence();
    }
    if (TAC_S3_DRAIN_LATEPUB && pend_val) asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"(pend_addr), "v"(pend_val) : "memory");
    D3_STAMP(3);
    D3_STAMP_FLUSH;
}


// =====================================================================================================================
// IN-PLACE form: no ring.  Frames are dealt statically (frame i of the workgroup's chunk belongs to transform wave i mod TW, its
// k-th frame is i = w + k TW); the finished row stays IN PLACE over the wave's own exchange area (like stft_stream3_kernel) and
// the wave goes on with its next frame's window and first butterfly — it needs the area back only for that frame's first
// exchange — so the depth between transform and stores is TW rows instead of what a ring finds room for (4 ... 8).
// What lets a row have ONE reader although spans are cut at 1 KB boundaries of the output: the end of row i that belongs to the
// first kilobyte of span i + 1 (the floats behind B_i+1, at most 255) is ALSO written to the wave's 1 KB tail buffer, and the
// producer of row i + 1 adds the <= 3 floats of its own head up to the next 16-byte boundary behind them: span j reads
// [B_j, seam_j) from the tail buffer of row j - 1 and [seam_j, B_j+1) from the area of row j, all in aligned 16-byte pieces.
// State: prod[w] = rows published by transform wave w; freed[w] = rows of wave w whose span has been read (written by the
// drain wave that read it; it also frees the tail buffer of the row before).  No atomics.
template <int NC, int E, int MODE, int TW, int DW>
struct Drain3iCfg {
    using F = WaveFft<NC, E>;
    static constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);
    static constexpr int XA = s3_xa_bytes<F>();
    static constexpr int TABLES = ST_TW_BYTES + 64 + 64 * (F::NPAIR + E) * (int)sizeof(cf);
    static constexpr int CTRL = 128;                                         // prod[16], freed[16]
    static constexpr int TB = 260;                                           // floats per tail buffer (255 + 3, 16-byte multiple)
    static constexpr int BYTES = TW * XA + TABLES + CTRL + TW * TB * 4;
    static constexpr int NSTMAX = (LENF + 255) / 256 + 1;
    static_assert(TW >= 2 && TW <= 16 && BYTES <= 160 * 1024 && XA >= (LENF + 3) * 4, "LDS");
};

template <int NC, int E, int MODE, int TW, int DW>
__global__ void __launch_bounds__((TW + DW) * 64, (TW + DW) / 4)
stft_drain3i_kernel(FrameGeom g, Tables tb, StftEpilogue ep, Stream3Launch lp) {
    using F = WaveFft<NC, E>;
    using D = Drain3iCfg<NC, E, MODE, TW, DW>;
    static_assert(F::G == 1 && E == 16 && radix_at(NC, 0) == 16, "fft_length 2048");
    
    constexpr int WAVES = TW + DW;
    constexpr int XA_BYTES = D::XA;
    constexpr int LENF = D::LENF;
    constexpr int TB = D::TB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const twlds = reinterpret_cast<float*>(smem_raw + (size_t)TW * XA_BYTES);
    cf* const ptwl = reinterpret_cast<cf*>(twlds + ST_TW_BYTES / 4 + 16);
    cf* const winl = ptwl + 64 * F::NPAIR;
    unsigned* const prod = reinterpret_cast<unsigned*>(winl + 64 * E);
    unsigned* const freed = prod + 16;
    float* const tails = reinterpret_cast<float*>(prod + D::CTRL / 4);
    const float half = 0.5f * g.scale;
    S3Setup<F, WAVES * 64> setup;
    setup.issue(g, tb, tid);

    const long long total = g.rows * g.n_frames;
    const long long chunk = lp.chunk;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < total ? begin + chunk : total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    const unsigned T = (unsigned)g.n_frames;
    const long long X0 = (long long)(reinterpret_cast<unsigned long long>(ep.out) >> 2) + begin * LENF;
    const unsigned prod_addr = lds_offset_of(prod), freed_addr = lds_offset_of(freed);
    typedef float f4 __attribute__((ext_vector_type(4)));

    if (w >= TW) {
        // =================================================== drain waves
        if (tid - TW * 64 < D::CTRL / 4) prod[tid - TW * 64] = 0u;
        setup.store(twlds, ptwl, winl, half, tid);
        __syncthreads();
        const int d = w - TW;
        if (TAC_S3_DRAIN_PRIO) __builtin_amdgcn_s_setprio(TAC_S3_DRAIN_PRIO);
        D3_STAMP_INIT;
        for (int j = d; j < nloc; j += DW) {
            D3_STAMP(3);                                      // store issue (+ loop overhead)
            const int wB = j % TW, kB = j / TW;
            const long long Xj = X0 + (long long)j * LENF;
            const long long lo = j == 0 ? X0 : (Xj & ~255LL);
            const long long hi = j == nloc - 1 ? Xj + LENF : ((Xj + LENF) & ~255LL);
            const long long lo16 = (lo + 3) & ~3LL;
            const int npre = (int)(lo16 - lo);                     // (first span only)
            const int nbody = (int)(hi - lo16);
            const int nch = nbody >> 2, ntail = nbody & 3;        // (tail floats: last span only)
            const int S = (int)(lo16 >> 2) & 63;
            const int NI = (nch + S + 63) >> 6;
            // chunks [0, cseam) of the body come from the tail buffer of row j - 1, the rest from the area of row j
            const int cseam = j > 0 ? (int)((((Xj + 3) & ~3LL) - lo16) >> 2) : 0;
            // float x of the span (x - lo16 = 4 c + e) is tbuf[4 c + e] below the seam, area[aoff + 4 c + e] from the seam on
            const float* const area = reinterpret_cast<const float*>(smem_raw + (size_t)wB * XA_BYTES);
            const int aoff = (int)(Xj & 3) - (int)(Xj - lo16);
            const float* const tbuf = tails + (j > 0 ? (j - 1) % TW : 0) * TB;
            if (j > 0) drain3_wait2(prod_addr + 4u * (unsigned)wB, (unsigned)kB + 1u, prod_addr + 4u * (unsigned)((j - 1) % TW), (unsigned)((j - 1) / TW) + 1u);
            else drain3_wait(prod_addr + 4u * (unsigned)wB, (unsigned)kB + 1u);
            D3_STAMP(0);                                      // waiting for the rows
            float hv = 0.0f, tv = 0.0f;
            if (npre) hv = area[aoff + (t < npre ? t : 0) - npre];
            if (ntail) tv = area[aoff + 4 * nch + (t < ntail ? t : 0)];
            f4 b[D::NSTMAX];
#pragma unroll
            for (int u = 0; u < D::NSTMAX; ++u) {
                if (u < NI) {
                    int c = t + 64 * u - S;
                    c = c < 0 ? 0 : (c < nch ? c : nch - 1);
                    const int ai = aoff + 4 * c;
                    const float* src = (u == 0 && c < cseam) ? tbuf + 4 * c : area + (ai < 0 ? 0 : ai);    // (the seam lies in the span's first kilobyte)
                    b[u] = *reinterpret_cast<const f4*>(src);
                }
            }
            {
                const unsigned fa = freed_addr + 4u * (unsigned)wB, fv = (unsigned)kB + 1u;
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"(fa), "v"(fv) : "memory");
            }
            D3_STAMP(1);                                      // span reads
            __builtin_amdgcn_sched_barrier(0);
#if TAC_S3_DRAIN_ABL == 1
            if (hv + tv + b[0].x != 12345.678f) continue;
#endif
            float* const gp = reinterpret_cast<float*>(static_cast<unsigned long long>(lo) << 2);
            if (npre && t < npre) gp[t] = hv;
            f4* const g4 = reinterpret_cast<f4*>(gp + npre);
#pragma unroll
            for (int u = 0; u < D::NSTMAX; ++u) {
                if (u < NI) {
                    const int c = t + 64 * u - S;
                    if (c >= 0 && c < nch) {
                        if (lp.plain_stores) g4[c] = b[u];
                        else __builtin_nontemporal_store(b[u], &g4[c]);
                    }
                }
            }
            if (ntail && t < ntail) gp[npre + 4 * nch + t] = tv;
        }
        D3_STAMP(3);
        D3_STAMP_FLUSH;
        return;
    }

    // ======================================================= transform waves
    cf* const xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);
    cf tw2[3];
    {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw2[q] = all[twiddles_before(NC, E, 2) + q];
    }
    cf v[E];
    int mode = 0, row = 0;
    long long fr = 0;
    constexpr bool EARLY = WAVES == 12;
    cf nx[EARLY ? E : 1];
    int nmode = 0, nrow = 0;
    long long nfr = 0;
    auto request_into = [&](int i, cf* dst, int& mode_o, int& row_o, long long& fr_o) {
        i = i < nloc ? i : nloc - 1;
        const unsigned gf = (unsigned)(begin + i);
        const unsigned r = gf / T;
        row_o = (int)r;
        fr_o = (long long)(gf - r * T);
        const long long start = fr_o * (long long)g.hop - g.center_pad;
        const bool ok = g.vec2_ok && start >= 0 && start + F::N <= g.length;
        mode_o = ok ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + F::N <= g.length ? cs : g.length - F::N;
        const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)row_o * g.row_stride + cs);
#pragma unroll
        for (int q = 0; q < E; ++q) dst[q] = src[t + q * F::LPF];
    };
    auto request = [&](int i) {
        if constexpr (EARLY) request_into(i, nx, nmode, nrow, nfr);
        else request_into(i, v, mode, row, fr);
    };
    if (nloc > 0) request(w);
    __builtin_amdgcn_sched_barrier(0);
    setup.store(twlds, ptwl, winl, half, tid);
    __syncthreads();
    if (nloc <= 0) return;

    S3Swz swz;
    swz.init(xa, t);
    const int nb = w + 1 < TW ? w + 1 : 0;                         // the wave of the row behind mine
    float* const tb_own = tails + w * TB;
    float* const tb_prev = tails + (w ? w - 1 : TW - 1) * TB;
    int k = 0;
    D3_STAMP_INIT;
    for (int i = w; i < nloc; i += TW, ++k) {
#if TAC_S3_DRAIN_STAMPS
        D3_STAMP(3);                                          // (row writes, publish, loop)
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
        D3_STAMP(0);                                          // waiting for the samples
#endif
        if constexpr (EARLY) {
            mode = nmode;
            row = nrow;
            fr = nfr;
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = nx[q];
        }
        // ---- s0: window, pass 0, exchange
        if (mode != 1) {
            if (k) drain3_wait(freed_addr + 4u * (unsigned)w, (unsigned)k);       // (the gather path stages its samples in the area)
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v, g, nullptr, xa, row, fr, tz, FetchF32{g.wave});
        }
#if TAC_S3_DRAIN_ABL == 2
        cf zm[F::NPAIR], zmid = v[0], ptw[F::NPAIR];
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            zm[p] = v[F::NPAIR + p];
            ptw[p] = mkc(1.0f, 0.0f);
        }
        if constexpr (EARLY) {
            __builtin_amdgcn_sched_barrier(0);
            request(i + TW);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k) drain3_wait(freed_addr + 4u * (unsigned)w, (unsigned)k);
#else
        {
            cf win[E];
            const f4* wl = reinterpret_cast<const f4*>(winl) + t;
#pragma unroll
            for (int u = 0; u < E / 2; ++u) {
                const f4 x = wl[u * 64];
                win[2 * u] = mkc(x.x, x.y);
                win[2 * u + 1] = mkc(x.z, x.w);
            }
            Dft<16>::run_windowed(v, win);
        }
        if constexpr (EARLY) {
            __builtin_amdgcn_sched_barrier(0);
            request(i + TW);
            __builtin_amdgcn_sched_barrier(0);
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
        // the area still holds my previous row until its span has been read
        D3_STAMP(4);                                          // window, first butterfly
        if (k) drain3_wait(freed_addr + 4u * (unsigned)w, (unsigned)k);
        D3_STAMP(1);                                          // waiting for the area
        s3_write_pass0_swz(v, swz);
        wave_lds_fence();
        s3_readback_pass1_swz(v, swz);
        // ---- s12
        F::template pass_twiddle<1, true>(v, tw1);
        F::template pass_butterflies<1>(v);
        F::exchange_1_2_in_registers(v);
        F::template pass_twiddle<2, true>(v, tw2);
        F::template pass_butterflies<2>(v);
        cf zm[F::NPAIR], zmid;
        s3_r2c_partners<F>(v, xa, zm, zmid, t);
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
        // ---- s3: R2C split; the row overwrites the exchange area once every lane holds its partners; its end goes to the tail
        //      buffer as well, its <= 3 head floats behind the end of the previous row in that row's tail buffer
        const long long Xi = X0 + (long long)i * LENF;
        const int a = (int)(Xi & 3);
        float* const stage = reinterpret_cast<float*>(xa) + a;
        const int ntl = (int)((Xi + LENF) & 255);                   // floats of this row's end in the first kilobyte of span i + 1
        const int hoff = (int)(Xi & 255);                           // where my head goes in the previous row's tail buffer
        // my tail buffer was last read by the span of the row behind my previous one
        D3_STAMP(5);                                          // exchange, passes 1 and 2, partners
        {
            const unsigned need = w + 1 < TW ? (unsigned)k : (k ? (unsigned)k + 1u : 0u);
            if (need) drain3_wait(freed_addr + 4u * (unsigned)nb, need);
        }
        D3_STAMP(2);                                          // waiting for the tail buffer
        if constexpr (MODE == 0) {
            cf xlo[F::NPAIR], xhi[F::NPAIR], xm, unused;
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) F::r2c_split_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], xlo[p], xhi[p]);
            F::r2c_split_x2(zmid, zmid, mkc(0.0f, -1.0f), xm, unused);
            wave_lds_fence();
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int kk = t + p * F::LPF;
                reinterpret_cast<cf*>(stage)[kk] = xlo[p];
                reinterpret_cast<cf*>(stage)[NC - kk] = xhi[p];
                if (p < 2 && kk < (ntl >> 1)) reinterpret_cast<cf*>(tb_own)[(ntl >> 1) - 1 - kk] = xhi[p];
            }
            if (t == 0) reinterpret_cast<cf*>(stage)[NC / 2] = xm;
            if (t == 0 && i > 0 && a == 2) reinterpret_cast<cf*>(tb_prev)[hoff >> 1] = xlo[0];
        } else {
            cf pw[F::NPAIR];
#pragma unroll
            for (int p = 0; p < F::NPAIR; p += 2)
                r2c_power_pair_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], v[F::reg_of_spectrum(p + 1)], zm[p + 1], ptw[p + 1], pw[p], pw[p + 1]);
            const float pmid = 4.0f * cnorm2(zmid);
            wave_lds_fence();
            const int npre = (4 - a) & 3;
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int kk = t + p * F::LPF;
                const float lov = spectral_row_value<MODE>(pw[p].x, ep), hiv = spectral_row_value<MODE>(pw[p].y, ep);
                stage[kk] = lov;
                stage[NC - kk] = hiv;
                if (p < 4 && kk < ntl) tb_own[ntl - 1 - kk] = hiv;
                if (p == 0 && i > 0 && t < npre) tb_prev[hoff + t] = lov;
            }
            if (t == 0) stage[NC / 2] = spectral_row_value<MODE>(pmid, ep);
        }
        if constexpr (!EARLY) {
            __builtin_amdgcn_sched_barrier(0);
            request(i + TW);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const unsigned pa = prod_addr + 4u * (unsigned)w, val = (unsigned)k + 1u;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"(pa), "v"(val) : "memory");
        }
        wave_lds_fence();
    }
    D3_STAMP(3);
    D3_STAMP_FLUSH;
}

}  // namespace tac
