// stft_stream3.hpp — the fft_length-2048 STFT / spectrogram rows with THREE (complex rows) or FOUR (real rows) waves per SIMD.
//
// Same front end as melspec_stream3_kernel (one frame per wave, twelve <=168- or sixteen <=128-register waves per CU, window, pass-1 and
// R2C twiddles from LDS); where that kernel contracts the |X|^2 row with the mel bank, this one streams the row out:
// it is staged IN PLACE over the frame's exchange area at the 16-byte phase of its place in the output, and leaves as
// unconditional 16-byte nontemporal stores (stft_pipe_kernel's epilogue).  Replaces
// torch.stft (reference functional.py:36-38) [+ complex_norm (functional.py:58-72)] [+ amplitude_to_db (functional.py:291-296)].
#pragma once
#include "melspec_stream3.hpp"

namespace tac {

template <int NC, int E, int WAVES>
__host__ __device__ inline size_t stft_stream3_lds_bytes() {
    using F = WaveFft<NC, E>;
    size_t xa = (size_t)s3_xa_bytes<F>();
    return (size_t)WAVES * xa + ST_TW_BYTES + 64 + 64 * (F::NPAIR + E) * sizeof(cf);
}

// launch parameters the host works out once: frames per workgroup (ceil(total / blocks)) and the row-store policy
struct Stream3Launch {
    long long chunk;
    int plain_stores;       // 0: nontemporal row stores (keep a cache-resident input resident), 1: plain stores (see launch_pipe3)
};

// MODE: 0 complex rows, 1 |X|^2, 2 |X|, 3 |X|^2 in dB, 4 |X| in dB (spectral_row_value)
// WAVES: 12 or 16 per workgroup (= per CU): the row-store form needs ~114 registers, so FOUR waves per SIMD fit as well
template <int NC, int E, int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4)
stft_stream3_kernel(FrameGeom g, Tables tb, StftEpilogue ep, Stream3Launch lp) {
    using F = WaveFft<NC, E>;
    static_assert(F::G == 1 && E == 16 && radix_at(NC, 0) == 16, "fft_length 2048");
    constexpr int XA_BYTES = s3_xa_bytes<F>();
    constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);
    constexpr int NST = ((LENF >> 2) + 63) / 64;          // 16-byte wave-stores per output row
    static_assert(XA_BYTES >= (LENF + 3) * 4, "the staged row (any 16-byte phase) fits the exchange area");
    static_assert(NST == 4 || NST == 8 || NST == 5 || NST == 9, "row store");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* const xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);
    float* const twlds = reinterpret_cast<float*>(smem_raw + (size_t)WAVES * XA_BYTES);
    unsigned* const next_frame = reinterpret_cast<unsigned*>(twlds + ST_TW_BYTES / 4);
    cf* const ptwl = reinterpret_cast<cf*>(next_frame + 16);
    cf* const winl = ptwl + 64 * F::NPAIR;
    const float half = 0.5f * g.scale;                    // the R2C split returns 2X: folded into the window
    S3Setup<F, WAVES * 64> setup;                         // tables: every load in flight, then the LDS stores (melspec_stream3.hpp)
    setup.issue(g, tb, tid);

    // Frames are dealt round-robin over all waves of the grid (round 5) — local index i is frame (i / WAVES) G_W + slot WAVES + i %
    // WAVES, the slots of one XCD (blockIdx % 8) side by side — so that the grid writes ONE tight window of adjacent rows instead of
    // one stream per workgroup: complex rows -4.1 %, real rows -2.6 % same process (profiles/r05/ab/batch54; the store pattern alone:
    // tools/ubench/row_store_rate.hip).  lp.chunk is not used.
    const long long total = g.rows * g.n_frames;
    const unsigned GW = gridDim.x * WAVES;
    const unsigned gslot = (gridDim.x & 7u) == 0 ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const long long nfull = total / GW, rem = total - nfull * GW - (long long)gslot * WAVES;
    const int nloc = (int)(nfull * WAVES + (rem < 0 ? 0 : (rem > WAVES ? WAVES : rem)));
    const unsigned T = (unsigned)g.n_frames;

    cf tw2[3];
    {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw2[q] = all[twiddles_before(NC, E, 2) + q];
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    cf v[E];
    int mode = 0, row = 0;
    long long fr = 0;
    // the twelve-wave complex-row form requests the NEXT frame's samples into their own registers as soon as the current frame's are
    // consumed (a whole frame of cover instead of the row stores' issue time): -1.1 ... -1.4 % on the complex rows (same-process
    // A/B); the real rows measure best with sixteen waves and the late request
    constexpr bool EARLY = WAVES == 12 && MODE == 0;
    cf nx[EARLY ? E : 1];           // EARLY: the requested (next) frame's samples; v is the frame being transformed
    int nmode = 0, nrow = 0;
    long long nfr = 0;
    auto request_into = [&](int i, cf* dst, int& mode_o, int& row_o, long long& fr_o) {
        i = i < nloc ? i : nloc - 1;
        const unsigned gf = ((unsigned)i / WAVES) * GW + gslot * WAVES + (unsigned)i % WAVES;
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
    // the wave's first frame is requested behind the table loads and ahead of the LDS stores and the barrier: its HBM latency
    // runs behind the rest of the set-up
    if (nloc > 0) request(w);
    __builtin_amdgcn_sched_barrier(0);
    if (tid == 0) *next_frame = WAVES;
    setup.store(twlds, ptwl, winl, half, tid);
    __syncthreads();
    if (nloc <= 0) return;

    S3Swz swz;
    swz.init(xa, t);
    int i = w;
    while (i < nloc) {
        unsigned ask = 0;
        int i_next = 0;
        if (t == 0) ask = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (EARLY) {                              // the frame requested during the previous one becomes current
            mode = nmode;
            row = nrow;
            fr = nfr;
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = nx[q];
        }
        const long long g0 = ((long long)row * T + fr) * LENF;      // this frame's row in the frame-major output
        // ---- s0: window, pass 0, exchange
        if (mode != 1) {                                    // frames touching the padding gather their samples first
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v, g, nullptr, xa, row, fr, tz, FetchF32{g.wave});
        }
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
        if constexpr (EARLY) {                              // nx is consumed: the next frame's samples go out now
            i_next = (int)__builtin_amdgcn_readfirstlane(ask);
            __builtin_amdgcn_sched_barrier(0);
            request(i_next);
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
        // ---- s3: R2C split; the row overwrites the exchange area once every lane holds its partners
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
        const int a = (int)(g0 & 3);
        float* const stage = reinterpret_cast<float*>(xa) + a;      // LDS and global share their 16-byte phase
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
            }
            if (t == 0) reinterpret_cast<cf*>(stage)[NC / 2] = xm;
        } else {
            cf pw[F::NPAIR];
#pragma unroll
            for (int p = 0; p < F::NPAIR; p += 2)
                r2c_power_pair_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], v[F::reg_of_spectrum(p + 1)], zm[p + 1], ptw[p + 1], pw[p], pw[p + 1]);
            const float pmid = 4.0f * cnorm2(zmid);
            wave_lds_fence();
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int kk = t + p * F::LPF;
                stage[kk] = spectral_row_value<MODE>(pw[p].x, ep);
                stage[NC - kk] = spectral_row_value<MODE>(pw[p].y, ep);
            }
            if (t == 0) stage[NC / 2] = spectral_row_value<MODE>(pmid, ep);
        }
        wave_lds_fence();
        // ---- the next frame's samples are requested BEFORE this row's stores (gfx950 counts loads and stores in one
        //      in-order vmcnt: this way "my samples have landed" does not wait for the stores behind them)
        if constexpr (EARLY) {
            i = i_next;
        } else {
            i = (int)__builtin_amdgcn_readfirstlane(ask);
            __builtin_amdgcn_sched_barrier(0);
            request(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the row leaves as 1 + NST + 1 unconditional stores: out-of-range lanes repeat a neighbour's element.  (Stores shifted
        //      onto 128-byte boundaries lower the arithmetic-free floor of this access pattern, 0.148 vs 0.176-0.186 ms, but not
        //      the kernel: tools/ablation/README.md, rounds 4 and 5.)
        float* const gdst = ep.out + g0;
        const int npre = (4 - a) & 3;
        const int nchunks = (LENF - npre) >> 2;
        {
            const int hmax = (npre > 1 ? npre : 1) - 1;
            const int hi = t < hmax ? t : hmax;
            gdst[hi] = stage[hi];
        }
        {
            const f4* const s4 = reinterpret_cast<const f4*>(stage + npre);
            f4* const g4 = reinterpret_cast<f4*>(gdst + npre);
            const int last = nchunks - 1;
            f4 b[NST];
            int c[NST];
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int j = t + 64 * u;
                c[u] = j < last ? j : last;
                b[u] = s4[c[u]];
            }
            __builtin_amdgcn_sched_barrier(0);            // all LDS reads in flight before the first store issues
            if (lp.plain_stores) {
#pragma unroll
                for (int u = 0; u < NST; ++u) g4[c[u]] = b[u];
            } else {
#pragma unroll
                for (int u = 0; u < NST; ++u) __builtin_nontemporal_store(b[u], &g4[c[u]]);
            }
        }
        {
            const int r = LENF - npre - 4 * nchunks;
            const int rmax = (r > 1 ? r : 1) - 1;
            const int ti = LENF - 1 - (t < rmax ? t : rmax);
            gdst[ti] = stage[ti];
        }
        wave_lds_fence();                                 // the next frame's first-pass writes follow these reads
    }
}

}  // namespace tac
