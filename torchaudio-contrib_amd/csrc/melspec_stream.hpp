// melspec_stream.hpp — the fused Melspectrogram(+dB) chain (layers.py:307-381) for fft_length 2048 as a
// barrier-free stream of frames: every wave of a persistent workgroup carries frames from samples to mel-dB rows
// on its own, TWO frames at a time, half a frame apart.
//
// Why.  melspec_sparse_kernel runs its three phases (FFT -> band-sparse contraction -> dB rows) in lockstep between
// workgroup barriers: all eight waves butterfly together (VALU busy, LDS idle), exchange together (LDS busy, VALU
// idle) and contract together (LDS reads only).  Its counters say exactly that — VALU ~40 % and LDS ~37 % busy, taking
// turns, with each of the 2 waves per SIMD sitting out its own LDS round trips (profiles/r01).  Here nothing is
// shared between waves but the constant tables, so there is nothing to wait for:
//
//   * a wave interleaves the stages s0..s4 of two frames ("threads" A and B of the wave) in a fixed rotation; while one
//     frame's exchange travels through the LDS the wave issues the other frame's butterflies.  The two frames share
//     ONE exchange area: the LDS executes a wave's instructions in order and every read-back is issued directly behind
//     the writes it depends on, so a frame's data is already on its way to registers when the other frame's writes
//     arrive.  Samples are requested one whole frame ahead into a register buffer the two threads take turns with.
//   * s3 leaves the frame's |X|^2 row in a private row buffer; s4 contracts it with the filterbank one band per lane
//     (lane l owns bands l, 64 + l, ...: the weights sit in LDS as [step][lane] pairs, conflict-free, and the
//     row reads of neighbouring bands are neighbouring addresses), applies the dB epilogue and stores the row of
//     out[frame][0..M) as 256-byte runs straight from registers — no output tile, no hand-over, no s_barrier.
//
// Frames are numbered globally (row * T + frame): out[rows][T][M] is contiguous in that index, so a workgroup simply
// owns a contiguous range of global frames, dealt round-robin to its 16 threads.
#pragma once
#include "mel_common.hpp"


namespace tac {

#ifndef TAC_ST_WAVES
#define TAC_ST_WAVES 8
#endif
constexpr int ST_WAVES = TAC_ST_WAVES;
constexpr int ST_MAX_SLOTS = 4;              // bands per lane (n_mels <= 256)
constexpr int ST_TW_STRIDE = 36;              // floats between the 16 pass-1 twiddle sets in LDS (144 B: conflict-free b128)
constexpr int ST_TW_BYTES = 16 * ST_TW_STRIDE * 4;
constexpr int ST_FAST_STEPS0 = 4, ST_FAST_STEPS1 = 16;   // slot lengths of the two-slot layout the FAST2 kernel is unrolled for
constexpr int ST_FAST_STEPS1_SHORT = 14;                 // ... and its second instantiation (no band of slot 1 beyond 56 bins: the standard 128-band bank)
// FAST2 tuning, measured at cfg-2 (DESIGN.md §3.3 has the sweeps):
constexpr int ST_RIDE1 = 6;      // steps of slot 1 whose reads ride along with slot 0 (4 / 6 / 8 / 10 -> 0.1375 / 0.1361 / 0.1358 / 0.1376 ms)
constexpr int ST_BATCH = 4;      // steps per round trip of the rest of slot 1 (14-step layout: 4 + 4; 8 + 6 no longer fits the registers: 0.156 ms)
constexpr int ST_RIDE = ST_FAST_STEPS0 + ST_RIDE1;     // steps issued at the end of s3
// A thread's next samples are requested behind its s4 (one stage ahead); requesting them behind its s3 (three ahead)
// takes registers from the ride-along reads and measured 0.1376 vs 0.1354 ms.

struct StreamArgs {
    const float* wl;       // device: weights [sum of steps][64 lanes][4 consecutive bins]
    const int* lo;         // device: first bin (multiple of 4) of lane l's band in slot s at [s * 64 + l]
    int nslot;             // bands per lane = ceil(n_mels / 64)
    int steps[ST_MAX_SLOTS];   // four-tap steps of each slot's loop (the longest band of the slot, rounded up to 4 steps)
    int wtot;              // floats in wl = 256 * sum(steps)
    int n_mels;
    int db;
    float amin;
    float log10_ref;
    float* out;            // [rows * T][M]
    long long total;       // rows * T
    const void* samples;   // the waveform in its own sample format (SampleFormat; g.wave when float32)
    const float* lut;      // device float[256]: mu-law decode table for the coded formats
    long long chunk;       // frames per workgroup = ceil(total / blocks) (melspec_stream3_kernel; the host divides once)
    unsigned long long* probe;   // diagnostics (tac_debug_clock_probe): [2 * block] = {shader cycles, 100 MHz ticks} of wave 0's frame loop; or null
    int rev;               // cell 64 s + l is band n_mels - 1 - (64 s + l) (banks whose band count is not a multiple of 64: pack_lanes)
};
constexpr int ST_REV_MARK = 256;               // in info_host[2] of such a pack (below LM_MARK)

// sample formats of the frame load (tac_amd.h TAC_SAMPLES_*)
// (FMT_*: host_common.hpp)

template <int NC, int E>
struct StreamCfg {
    using F = WaveFft<NC, E>;
    static constexpr int NBINS = NC + 1;
    static constexpr int PROW0 = NBINS + 7;                                   // + slack for zero-weight taps past the row
    static constexpr int PROW = (PROW0 + 3) & ~3;                             // rows stay 16-byte aligned
    static_assert(F::G == 1, "one frame per wave-pass");
};

template <int NC, int E>
__host__ __device__ inline size_t stream_lds_bytes(int wtot) {
    using C = StreamCfg<NC, E>;
    size_t b = (size_t)ST_WAVES * ((C::F::PADDED * sizeof(cf) + 15) & ~(size_t)15);
    b += (size_t)ST_WAVES * 2 * C::PROW * 4;
    b += ((size_t)wtot * 4 + 15) & ~(size_t)15;
    return b + 1024 + ST_TW_BYTES + 16;                                        // mu-law decode table, pass-1 twiddles, frame counter
}

// FAST2: 0 = the general kernel, otherwise the number of steps of slot 1 in the (4, FAST2)-step specialisation
template <int NC, int E, bool POW2, bool FULLM, int FMT, int FAST2>
__global__ void __launch_bounds__(ST_WAVES * 64, 2)
melspec_stream_kernel(FrameGeom g, Tables tb, StreamArgs m) {
    using C = StreamCfg<NC, E>;
    using F = typename C::F;
    constexpr int PROW = C::PROW, NBINS = C::NBINS, SLOTS = 2 * ST_WAVES;
    constexpr int XA_BYTES = (F::PADDED * sizeof(cf) + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);                 // this wave's exchange area
    float* const rows_all = reinterpret_cast<float*>(smem_raw + (size_t)ST_WAVES * XA_BYTES);
    float* rowA = rows_all + (size_t)(2 * w) * PROW;                                 // |X|^2 row of thread A / B
    float* rowB = rowA + PROW;
    float* wlds = rows_all + (size_t)ST_WAVES * 2 * PROW;

    float* lutlds = wlds + ((m.wtot + 3) & ~3);                                      // mu-law decode table (coded inputs)
    for (int i = tid; i < m.wtot; i += ST_WAVES * 64) wlds[i] = m.wl[i];
    if (FMT >= FMT_MULAW_U8 && tid < 256) lutlds[tid] = m.lut[tid];
    float* twlds = lutlds + 256;                                                     // pass-1 twiddles by (lane & 15) (FAST2)
    if (FAST2 && tid < 16 * 16) {
        const int js = tid >> 4, q = tid & 15;                                       // q = 0 is a pad slot
        const cf wv = q ? tb.w_nc[js * q * (NC / 256)] : mkc(1.0f, 0.0f);
        twlds[js * ST_TW_STRIDE + 2 * (q ? q - 1 : 15)] = wv.x;
        twlds[js * ST_TW_STRIDE + 2 * (q ? q - 1 : 15) + 1] = wv.y;
    }
    unsigned* const next_frame = reinterpret_cast<unsigned*>(twlds + ST_TW_BYTES / 4);   // the workgroup's frame counter
    if (tid == 0) *next_frame = SLOTS;                                               // (the first SLOTS frames are dealt by wave number)
    for (int i = tid; i < ST_WAVES * 2 * (PROW - NBINS); i += ST_WAVES * 64) {       // slack columns stay zero for good
        const int r = i / (PROW - NBINS), c2 = i - r * (PROW - NBINS);
        rows_all[(size_t)r * PROW + NBINS + c2] = 0.0f;
    }

    // this workgroup's contiguous range of global frames
    const long long chunk = (m.total + gridDim.x - 1) / gridDim.x;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < m.total ? begin + chunk : m.total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    const unsigned T = (unsigned)g.n_frames;

    const int t = lane;
    // inter-pass twiddles: pass 2's three stay in registers; pass 1's fifteen too, except in the FAST2 kernel, which
    // re-reads them from LDS every frame (they are needed during s12 only) to make room for the contraction's
    // ride-along reads
    cf tw[FAST2 ? 3 : F::NTW];
    if constexpr (FAST2) {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw[q] = all[twiddles_before(NC, E, 2) + q];
    } else {
        F::load_twiddles(tw, tb.w_nc, t);
    }
    const cf w0 = tb.w_n[t];                                           // R2C: W_N^{t + 64 i} = W_N^t * W_32^i
    constexpr bool FULLPTW = FAST2 != 0;       // all eight R2C twiddles in registers (the general kernel has no room: one x constants)
    cf ptw[FULLPTW ? F::NPAIR : 1];
    if constexpr (FULLPTW) {
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) ptw[p] = tb.w_n[t + p * F::LPF];
    }
    cf win[E];
    load_window_regs<F>(win, g, t);
    // 2X -> scale * X once, in the window; int16 PCM samples are integers there, their 2^-15 goes in as well
    const float half = 0.5f * g.scale;
    const float win_scale = FMT == FMT_I16 ? half * (1.0f / 32768.0f) : half;
#pragma unroll
    for (int e = 0; e < E; ++e) win[e] = cscale(win[e], win_scale);
    int lo_s[ST_MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < ST_MAX_SLOTS; ++s) lo_s[s] = s < m.nslot ? m.lo[s * 64 + lane] : 0;
    __syncthreads();                                                   // the only barrier of the kernel (tables in place)

#define ST_MARK(k) do { } while (0)
    cf vA[E], vB[E];
    cf zmA[F::NPAIR], zmB[F::NPAIR], zmidA = mkc(0.f, 0.f), zmidB = mkc(0.f, 0.f);

    // Request the samples of local frame i straight into the thread's (by then dead) data registers, three stages
    // before they are used.  The sixteen loads are issued for EVERY frame — for a frame that touches the padding from a
    // clamped, in-range address (its data is gathered at consumption time instead, mode 2) — and indices past the end of
    // the range are clamped to its last frame, which is then simply computed and stored twice with identical values:
    // the instruction stream is the same on every path, so hipcc's waitcnt pass can count (gfx950 has ONE in-order
    // vmcnt for loads and stores; an uncounted wait for these loads would also wait for the row stores behind them).
    // The workgroup's frames are taken from a shared counter, not dealt out in fixed strides: the two waves of a SIMD do
    // not get equal shares of its issue slots (the older one wins the arbitration: measured 230 k vs 330 k cycles for the
    // same 39 frames), so with equal shares the SIMD would run its last quarter with one wave and nothing to overlap.
    // (two halves so that the counter's answer can travel with other LDS traffic: FAST2 asks at the end of s3 and
    // looks at the answer after s4)
    auto grab_ask = [&]() -> unsigned {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;
    };
    auto grab = [&]() -> int { return (int)__builtin_amdgcn_readfirstlane(grab_ask()); };
    auto request = [&](cf (&raw)[E], int i, int& mode, int& row, long long& fr) {
        i = i < nloc ? i : nloc - 1;
        const unsigned gf = (unsigned)(begin + i);
        const unsigned r = gf / T;
        row = (int)r;
        fr = (long long)(gf - r * T);
        const long long start = fr * (long long)g.hop - g.center_pad;
        const bool ok = g.vec2_ok && start >= 0 && start + F::N <= g.length;
        mode = ok ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + F::N <= g.length ? cs : g.length - F::N;             // host guarantees length >= N
        const long long off = (long long)row * g.row_stride + cs;      // in samples
        {
            if constexpr (FMT == FMT_F32) {
                const cf* src = reinterpret_cast<const cf*>(static_cast<const float*>(m.samples) + off);
#pragma unroll
                for (int q = 0; q < E; ++q) raw[q] = src[t + q * F::LPF];
            } else if constexpr (FMT == FMT_I16) {                      // a pair of samples = one dword
                const unsigned* src = reinterpret_cast<const unsigned*>(static_cast<const short*>(m.samples) + off);
#pragma unroll
                for (int q = 0; q < E; ++q) raw[q].x = __uint_as_float(src[t + q * F::LPF]);
            } else if constexpr (FMT == FMT_MULAW_U8) {                 // a pair of codes = one 16-bit load
                const unsigned short* src =
                    reinterpret_cast<const unsigned short*>(static_cast<const unsigned char*>(m.samples) + off);
#pragma unroll
                for (int q = 0; q < E; ++q) raw[q].x = __uint_as_float((unsigned)src[t + q * F::LPF]);
            } else {                                                    // int64 codes: the low dword of each
                const int* src = reinterpret_cast<const int*>(static_cast<const long long*>(m.samples) + off);
#pragma unroll
                for (int q = 0; q < E; ++q) {
                    raw[q].x = __int_as_float(src[4 * (t + q * F::LPF)]);
                    raw[q].y = __int_as_float(src[4 * (t + q * F::LPF) + 2]);
                }
            }
        }
    };
    // the requested registers as float sample pairs (still unwindowed): PCM integers / decoded codes
    auto decode = [&](cf (&v)[E]) {
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
    // sample access of the gather path (frames touching the padding), in the same units as `decode`
    struct Fetch {
        const void* base;
        const float* lut;
        __device__ __forceinline__ float operator()(long long row_offset, int j) const {
            if constexpr (FMT == FMT_F32) return static_cast<const float*>(base)[row_offset + j];
            else if constexpr (FMT == FMT_I16) return (float)static_cast<const short*>(base)[row_offset + j];
            else if constexpr (FMT == FMT_MULAW_U8) return lut[static_cast<const unsigned char*>(base)[row_offset + j]];
            else return lut[(unsigned)static_cast<const long long*>(base)[row_offset + j] & 0xffu];
        }
    };
    // s0: windowed samples -> pass-0 butterflies -> exchange 0 (write, read-back issued)
    auto s0 = [&](cf (&v)[E], int mode, int row, long long fr) {
        if (mode == 1) {
            decode(v);
        } else {                                                       // edge / unaligned frame: gathered through the exchange
            int tz;                                                    // (an opaque copy of the lane number: this rare path's
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));      // address registers must not be hoisted out of the loop)
            load_frame<F, true, true>(v, g, win, xa, row, fr, tz, Fetch{m.samples, lutlds});  // area, plain window ->
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = cscale(v[e], win_scale);                       // the scale goes here
        }
    };
    typedef float f4 __attribute__((ext_vector_type(4)));
    // (fast-path frames arrive unwindowed: the window is folded into pass 0's first butterflies, -8 packed instructions)
    auto s0b = [&](cf (&v)[E], int mode, auto&& before_writes) {
        if (mode == 1) Dft<16>::run_windowed(v, win);
        else F::template pass_butterflies<0>(v);
        wave_lds_fence();
        before_writes();
        F::template pass_write<0, true>(v, xa, t, t);
        wave_lds_fence();
        F::template pass_readback<1>(v, xa, t);
    };
    // FAST2: pass 1's twiddles are requested at the end of the stage that precedes the s12 they are for (one set of
    // registers serves both threads)
    auto tw1_issue = [&](cf (&tw1)[16]) {
        if constexpr (FAST2) {
            const f4* tl = reinterpret_cast<const f4*>(twlds + (t & 15) * ST_TW_STRIDE);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4 x = tl[u];
                tw1[2 * u] = mkc(x.x, x.y);
                tw1[2 * u + 1] = mkc(x.z, x.w);
            }
        }
    };
    // s12: pass 1, the in-register exchange, pass 2; the lower half of the spectrum stays in registers, the upper half
    // travels to its R2C partners
    auto s12 = [&](cf (&v)[E], cf (&zm)[F::NPAIR], cf& zmid, cf (&tw1)[16], bool reload_tw1) {
        if constexpr (FAST2) F::template pass_twiddle<1, true>(v, tw1);
        else F::template pass_twiddle<1>(v, tw);
        F::template pass_butterflies<1>(v);
        F::exchange_1_2_in_registers(v);
        if constexpr (FAST2) F::template pass_twiddle<2, true>(v, tw);
        else F::template pass_twiddle<2>(v, tw);
        F::template pass_butterflies<2>(v);
        wave_lds_fence();
        F::template pass_write<2, true>(v, xa, t, t);
        wave_lds_fence();
        // partners Z[NC - t - 64 p]: one address register, the rest are immediates (pad(a - c) = pad(a) - pad(c) for
        // multiples of 16); lane 0's first partner is Z[0] itself (its read lands one slot past the area and is dropped)
        const cf* const pb = xa + lds_pad(NC - t);
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            const cf z = pb[-lds_pad_c(p * F::LPF)];
            zm[p] = (p == 0 && t == 0) ? v[F::reg_of_spectrum(0)] : z;
        }
        zmid = xa[lds_pad(NC / 2)];
    };
    // s3: R2C split, |X|^p row into the thread's row buffer
    auto s3 = [&](cf (&v)[E], cf (&zm)[F::NPAIR], cf zmid, float* prow) {
        wave_lds_fence();
        static_assert(F::NPAIR % 2 == 0, "R2C pairs are processed two at a time");
#pragma unroll
        for (int p = 0; p < F::NPAIR; p += 2) {
            const int kk = t + p * F::LPF, kk2 = kk + F::LPF;
            cf pw, pw2;
            if constexpr (FULLPTW) {
                r2c_power_pair_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], v[F::reg_of_spectrum(p + 1)], zm[p + 1], ptw[p + 1], pw, pw2);
            } else {
                pw = F::r2c_power_factored_x2(v[F::reg_of_spectrum(p)], zm[p], w0, p);
                pw2 = F::r2c_power_factored_x2(v[F::reg_of_spectrum(p + 1)], zm[p + 1], w0, p + 1);
            }
            prow[kk] = POW2 ? pw.x : __builtin_amdgcn_sqrtf(pw.x);
            prow[NC - kk] = POW2 ? pw.y : __builtin_amdgcn_sqrtf(pw.y);
            prow[kk2] = POW2 ? pw2.x : __builtin_amdgcn_sqrtf(pw2.x);
            prow[NC - kk2] = POW2 ? pw2.y : __builtin_amdgcn_sqrtf(pw2.y);
        }
        if (t == 0) {
            const float pm = 4.0f * cnorm2(zmid);                       // X[NC/2] = conj(Z[NC/2]); Z carries the 0.5
            prow[NC / 2] = POW2 ? pm : __builtin_amdgcn_sqrtf(pm);
        }
        wave_lds_fence();
    };
    // FAST2 (two slots of 4 and 16 steps): the first ten steps' reads are issued right behind the row (end of s3) and
    // ride along with the other frame's next stage; s4 finds them landed.  The weights of those steps are read ONCE per
    // pair of frames (B's issue): the paired rotation runs B.s4 and A.s4 back to back and both use the same weights, so
    // A's issue (from inside B.s4) fetches its row only and A.s4 reuses B's weight registers.
    auto s3_issue = [&](const float* prow, f4 (&cw)[ST_RIDE], f4 (&cp)[ST_RIDE], bool with_weights) {
        const f4* wp = reinterpret_cast<const f4*>(wlds) + lane;
        const f4* p0 = reinterpret_cast<const f4*>(prow + lo_s[0]);
        const f4* p1 = reinterpret_cast<const f4*>(prow + lo_s[1]);
#pragma unroll
        for (int u = 0; u < ST_FAST_STEPS0; ++u) {
            if (with_weights) cw[u] = wp[u * 64];
            cp[u] = p0[u];
        }
#pragma unroll
        for (int u = 0; u < ST_RIDE1; ++u) {
            if (with_weights) cw[ST_FAST_STEPS0 + u] = wp[(ST_FAST_STEPS0 + u) * 64];
            cp[ST_FAST_STEPS0 + u] = p1[u];
        }
    };
    // s4: filterbank contraction (one band per lane and slot; four taps per step: one 16-byte weight read, one 16-byte
    // row read, two packed FMAs), dB, row store.  A trip is eight steps whose sixteen reads are all issued before the
    // first FMA (the thread's data registers are free by now), so a slot costs one or two LDS round trips, not one per
    // step; slot lengths are whole half-trips (tac_melbank_pack).
    auto fma4 = [](f4 wv, f4 pv, cf& a0, cf& a1) {
        a0 = __builtin_elementwise_fma(mkc(wv.x, wv.y), mkc(pv.x, pv.y), a0);
        a1 = __builtin_elementwise_fma(mkc(wv.z, wv.w), mkc(pv.z, pv.w), a1);
    };
    const bool fast_db = m.amin >= 1.1754944e-38f;                      // (uniform) hardware log2 unless the clamp admits denormals
    const float ten_log10_ref = 10.0f * m.log10_ref;
    // `after_ride` runs once the ride-along steps are consumed (their registers are free again): the paired rotation
    // issues the other thread's ride-along reads there.  The first batch of the rest is requested before the ride-along
    // steps are consumed, so its round trip overlaps their FMAs.
    auto s4_fast = [&](const float* prow, int i, const f4 (&cw)[ST_RIDE], const f4 (&cp)[ST_RIDE], auto&& after_ride) {
        i = i < nloc ? i : nloc - 1;
        constexpr int REST = (FAST2 ? FAST2 : ST_FAST_STEPS1) - ST_RIDE1;
        constexpr int NB = (REST + ST_BATCH - 1) / ST_BATCH;
        const f4* wp = reinterpret_cast<const f4*>(wlds) + lane + ST_RIDE * 64;
        const f4* p1 = reinterpret_cast<const f4*>(prow + lo_s[1]) + ST_RIDE1;
        cf a0 = mkc(0.f, 0.f), a1 = mkc(0.f, 0.f), b0 = mkc(0.f, 0.f), b1 = mkc(0.f, 0.f);
        f4 w2[ST_BATCH], q2[ST_BATCH];
        auto issue_batch = [&](int c0) {
#pragma unroll
            for (int u = 0; u < ST_BATCH; ++u) {
                if (c0 + u < REST) {
                    w2[u] = wp[(c0 + u) * 64];
                    q2[u] = p1[c0 + u];
                }
            }
        };
        if constexpr (NB > 0) issue_batch(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < ST_FAST_STEPS0; ++u) fma4(cw[u], cp[u], a0, a1);
#pragma unroll
        for (int u = 0; u < ST_RIDE1; ++u) fma4(cw[ST_FAST_STEPS0 + u], cp[ST_FAST_STEPS0 + u], b0, b1);
        // the FMAs above must really be done (their operands' registers free) before after_ride's reads are issued:
        // sched_barrier orders machine instructions only, LLVM's IR passes would sink the FMAs below the reads
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
        after_ride();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NB; ++c) {                                  // the remaining steps of slot 1, ST_BATCH per round trip
#pragma unroll
            for (int u = 0; u < ST_BATCH; ++u)
                if (c * ST_BATCH + u < REST) fma4(w2[u], q2[u], b0, b1);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < NB) issue_batch((c + 1) * ST_BATCH);
            __builtin_amdgcn_sched_barrier(0);
        }
        float v0 = (a0.x + a0.y) + (a1.x + a1.y), v1 = (b0.x + b0.y) + (b1.x + b1.y);
        if (m.db) {
            v0 = fast_db ? amp_to_db_fast(v0, m.amin, ten_log10_ref) : amp_to_db(v0, m.amin, m.log10_ref);
            v1 = fast_db ? amp_to_db_fast(v1, m.amin, ten_log10_ref) : amp_to_db(v1, m.amin, m.log10_ref);
        }
        float* orow = m.out + (begin + i) * (long long)m.n_mels + lane;
        orow[0] = v0;
        orow[64] = v1;
    };
    auto s4 = [&](const float* prow, int i) {
        i = i < nloc ? i : nloc - 1;
        const f4* wp = reinterpret_cast<const f4*>(wlds) + lane;
        float* orow = m.out + (begin + i) * (long long)m.n_mels + (m.rev ? m.n_mels - 1 - lane : lane);
        const int ostep = m.rev ? -64 : 64;                               // cell 64 s + lane -> its band (StreamArgs::rev)
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
                if (j + 4 <= n) {                                         // half trip
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
                for (; j < n; ++j, wp += 64) fma4(wp[0], pp[j], acc0, acc1);   // (slot lengths that are not whole half-trips: the 14-step layout)
                float v = (acc0.x + acc0.y) + (acc1.x + acc1.y);
                if (m.db) v = fast_db ? amp_to_db_fast(v, m.amin, ten_log10_ref) : amp_to_db(v, m.amin, m.log10_ref);
                if (FULLM || s * 64 + lane < m.n_mels) orow[s * ostep] = v;
            }
        }
    };

    // First frames of thread A: 2w, of thread B: 2w + 1, then whatever the counter hands out.  Rotation of one iteration:
    //   A.s0 | B.s12 | A.s12 | B.s3 + request | A.s3 + request | B.s4 | A.s4 | B.s0
    // so every stage's LDS round trip was issued one other-thread stage earlier, and the FFT stages (the register peak)
    // never run while contraction reads are in flight.  FAST2: B.s3 ends by issuing B's first contraction reads (they
    // ride along with A.s3), B.s4 consumes them and then issues A's (which ride along with the rest of B.s4); the sample
    // request follows s4.  B's first frame is brought to the point behind s0 before the loop.  No branch
    // inside the loop: a wave stops when neither of its threads holds a frame of the range; a thread without one
    // recomputes the last frame (request clamps) and stores identical values.
    if (nloc > 0) {
        int modeA, rowA_, modeB, rowB_;
        long long frA, frB;
        int iA = 2 * w, iB = 2 * w + 1, nA = 0, nB = 0;
        unsigned askA = 0, askB = 0;
        cf tw1[16];
        f4 cw[ST_RIDE], cpA[ST_RIDE], cpB[ST_RIDE];
        request(vB, iB, modeB, rowB_, frB);
        request(vA, iA, modeA, rowA_, frA);
        s0(vB, modeB, rowB_, frB);
        s0b(vB, modeB, []() {});
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): the loop is entered with nothing in flight
        ST_MARK(6);
#pragma unroll 1
        while (iA < nloc || iB < nloc) {
            s0(vA, modeA, rowA_, frA);
            s0b(vA, modeA, []() {});
            tw1_issue(tw1);
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(1);
            s12(vB, zmB, zmidB, tw1, true);
            tw1_issue(tw1);
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(2);
            s12(vA, zmA, zmidA, tw1, false);
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(2);
            s3(vB, zmB, zmidB, rowB);
            if constexpr (FAST2) {
                s3_issue(rowB, cw, cpB, true);
                askB = grab_ask();
            } else {
                nB = grab();
                request(vB, nB, modeB, rowB_, frB);
            }
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(4);
            s3(vA, zmA, zmidA, rowA);
            if constexpr (!FAST2) {
                nA = grab();
                request(vA, nA, modeA, rowA_, frA);
            }
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(4);
            if constexpr (FAST2) {
                s4_fast(rowB, iB, cw, cpB, [&]() {
                    s3_issue(rowA, cw, cpA, false);
                    askA = grab_ask();
                });
                iB = (int)__builtin_amdgcn_readfirstlane(askB);
                request(vB, iB, modeB, rowB_, frB);                     // the contraction used the frame's registers until here
            } else {
                s4(rowB, iB);
                iB = nB;
            }
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(5);
            if constexpr (FAST2) {
                s4_fast(rowA, iA, cw, cpA, []() {});
                iA = (int)__builtin_amdgcn_readfirstlane(askA);
                request(vA, iA, modeA, rowA_, frA);
            } else {
                s4(rowA, iA);
                iA = nA;
            }
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(5);
            s0(vB, modeB, rowB_, frB);
            s0b(vB, modeB, []() {});
            __builtin_amdgcn_sched_barrier(0);
            ST_MARK(1);
        }
    }
}

}  // namespace tac
