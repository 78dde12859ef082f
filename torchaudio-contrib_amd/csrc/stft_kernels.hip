// stft_kernels.hip — framed, windowed, batched R2C FFT with complex / |.|^p (/dB) epilogues.
//
// Replaces torch.stft (+ torch.norm/pow, + amplitude_to_db) on the reference path
// (torchaudio_contrib/functional.py:99-107, 126-128, 291-296).  One wave = G frames; each
// 4-wave workgroup walks a contiguous chunk of (row, frame) groups so the 4x overlap between
// consecutive frames (hop = N/4) is served by L1/L2 and every input sample leaves HBM once.
// Reflect / constant / replicate / circular padding is done by index arithmetic on the unpadded
// waveform — no padded copy, no framed copy, no separate window multiply.
#include "host_common.hpp"
#include "stft_stream3.hpp"
#include "stft_ring3.hpp"

#ifndef TAC_S3_RING_TW
#define TAC_S3_RING_TW 12         // transform waves of the hop-ring form (+ one loader wave)
#endif



namespace tac {

using StftStamp = NoStamp;

constexpr int STFT_WAVES = 4;


// |X|^p (+ dB) for the general case, kept out of the unrolled per-bin code: the squared magnitudes are
// parked in the frame's own LDS buffer and transformed by this ROLLED loop (one copy of powf/log10f in
// the kernel instead of 34 per frame), which also turns the stores into full coalesced rows.
template <int NC, int LPF>
__device__ __noinline__ void finish_power_row(const float* prow, float* obase, int t, int nbins, float power,
                                              int db, float amin, float log10_ref) {
#pragma unroll 1
    for (int bin = t; bin < nbins; bin += LPF) {
        float s = prow[bin <= NC ? bin : 2 * NC - bin];
        float v = (power == 2.0f) ? s : ((power == 1.0f) ? sqrtf(s) : powf(sqrtf(s), power));
        if (db) v = amp_to_db(v, amin, log10_ref);
        obase[bin] = v;
    }
}

// NF frames per wave can be advanced together (see WaveFft::run); with the inter-pass and R2C twiddles held
// in registers the kernel sits at 2 waves/SIMD.  NF = 1 is what ships (see launch_stft).
template <int NC, int E, int MODE, int NF, bool HOIST, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 2)
stft_kernel(FrameGeom g, Tables tb, StftEpilogue ep) {
    using F = WaveFft<NC, E>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* smem = reinterpret_cast<cf*>(smem_raw);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / F::LPF;
    const int t = lane % F::LPF;
    constexpr int WAVE_SLOTS = ((NF * F::G * F::PADDED + 1) / 2) * 2;     // complex slots per wave, 16-byte multiple
    cf* lds[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) lds[f] = smem + w * WAVE_SLOTS + (f * F::G + sub) * F::PADDED;

    cf tw[F::NTW];
    cf ptw[HOIST ? F::NPAIR : 1];
    if constexpr (HOIST) {
        F::load_twiddles(tw, tb.w_nc, t);
#pragma unroll
        for (int i = 0; i < F::NPAIR; ++i) ptw[i] = tb.w_n[t + i * F::LPF];
    }

    // work unit of one wave-iteration: NF*G consecutive frames of ONE row (so their output rows are adjacent)
    constexpr int FPU = NF * F::G;
    const int units_per_row = (int)((g.n_frames + FPU - 1) / FPU);
    const int total = (int)g.rows * units_per_row;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total ? begin + chunk : total;
    const int nbins = ep.onesided ? NC + 1 : 2 * NC;
    const long long per_frame = (long long)nbins * (MODE == 0 ? 2 : 1);
    const float hscale = 0.5f * g.scale;                  // the R2C split returns 2·X

    StftStamp st;
    // units are taken from a workgroup counter (behind the wave slots), not dealt out in fixed strides: the older of two
    // waves that share a SIMD wins the issue arbitration and would finish its share long before the other
    unsigned* const next_unit = reinterpret_cast<unsigned*>(smem + WAVES * WAVE_SLOTS);
    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + WAVES);
    __syncthreads();
    for (int unit = begin + w; unit < end;) {
        st.mark(0);                                         // loop overhead + previous iteration's store issue
        int nxt_unit;
        {
            unsigned nv = 0;
            if (lane == 0) nv = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nxt_unit = (int)__builtin_amdgcn_readfirstlane(nv);
        }
        const int urow = unit / units_per_row;
        const long long uframe0 = (long long)(unit - urow * units_per_row) * FPU;
        cf v[NF][E];
        long long row[NF], frame[NF];
        int tl = t;
        asm volatile("" : "+v"(tl));          // launder: keeps the per-iteration table loads inside the loop
        cf win[F::E];                     // window: L1-resident, shared by the NF frames of this iteration
        load_window_regs<F>(win, g, tl);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            row[f] = urow;
            frame[f] = uframe0 + f * F::G + sub;            // may be >= T: load_frame() then yields zeros
            load_frame<F, true>(v[f], g, win, lds[f], row[f], frame[f], t);
        }
        if constexpr (!HOIST) {
            // twiddles only once the frame is windowed (the window's registers are free by then): N = 4096 keeps
            // 64 data registers per lane and spilled 260 B with the tables requested up front
            __builtin_amdgcn_sched_barrier(0);
            F::load_twiddles(tw, tb.w_nc, tl);
        }
        st.mark(8);                                         // frame loaded (global latency) and windowed
        F::template run<NF>(v, lds, tw, t, st);
        st.mark(9);                                         // last pass's spectrum written to LDS
        const bool simple = (MODE == 0) ? (ep.onesided != 0) : (ep.onesided && ep.power == 2.0f && !ep.db);
        bool done = false;
        if constexpr (F::G == 1) {
            if (simple) {
                // Wide-store epilogue: the unit's NF output rows are adjacent in memory, so they are staged in
                // output order in LDS (in place over the consumed spectra) and streamed out with 16-byte
                // ds_read_b128 -> global_store_dwordx4, ~8 store instructions per frame instead of 34 narrow ones
                // (the narrow stores were issue-bound: 0.22 of 0.44 ms at cfg-2).  The staging origin is shifted
                // by the rows' misalignment so LDS and global addresses share their 16-byte phase.
    constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);
                const long long g0 = ((long long)urow * g.n_frames + uframe0) * LENF;
                const int a = (int)(g0 & 3);
                float* stage = reinterpret_cast<float*>(lds[0]) + a;
                int nlive = 0;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    if (frame[f] < g.n_frames) {                                  // wave-uniform
                        cf xa[F::NPAIR], xb[F::NPAIR], xm, unused;
#pragma unroll
                        for (int i = 0; i < F::NPAIR; ++i) {
                            F::r2c_pair(lds[f], t + i * F::LPF, HOIST ? ptw[i] : tb.w_n[tl + i * F::LPF], xa[i], xb[i]);
                            xa[i] = cscale(xa[i], hscale); xb[i] = cscale(xb[i], hscale);
                        }
                        F::r2c_pair(lds[f], NC / 2, mkc(0.0f, -1.0f), xm, unused);
                        xm = cscale(xm, hscale);
                        wave_lds_fence();                                         // every Z of this frame is in registers
                        float* srow = stage + f * LENF;
#pragma unroll
                        for (int i = 0; i < F::NPAIR; ++i) {
                            const int k = t + i * F::LPF;
                            if constexpr (MODE == 0) {
                                reinterpret_cast<cf*>(srow)[k] = xa[i];
                                reinterpret_cast<cf*>(srow)[NC - k] = xb[i];
                            } else {
                                srow[k] = cnorm2(xa[i]);
                                srow[NC - k] = cnorm2(xb[i]);
                            }
                        }
                        if (t == 0) {
                            if constexpr (MODE == 0) reinterpret_cast<cf*>(srow)[NC / 2] = xm;
                            else srow[NC / 2] = cnorm2(xm);
                        }
                        wave_lds_fence();
                        ++nlive;
                    }
                }
                st.mark(10);                                // R2C split + staging of the output row
                const int len = nlive * LENF;
                float* gdst = ep.out + g0;
                const int npre = (4 - a) & 3;
                if (lane < npre && lane < len) gdst[lane] = stage[lane];
                const int nchunks = len > npre ? (len - npre) >> 2 : 0;
                const float4* s4 = reinterpret_cast<const float4*>(stage + npre);
                float4* g4 = reinterpret_cast<float4*>(gdst + npre);
#pragma unroll 4
                for (int c = lane; c < nchunks; c += 64) g4[c] = s4[c];
                const int tail0 = npre + 4 * nchunks;
                if (lane < len - tail0) gdst[tail0 + lane] = stage[tail0 + lane];
                done = true;
            }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (done) break;
            const bool live = frame[f] < g.n_frames;
            float* obase = ep.out + (row[f] * g.n_frames + (live ? frame[f] : 0)) * per_frame;
            cf* o2 = reinterpret_cast<cf*>(obase);
            float* prow = reinterpret_cast<float*>(lds[f]);
            if (simple) {
                // common case: X (complex) or |X|^2 goes straight from registers to coalesced stores
                if (live) {
#pragma unroll
                    for (int i = 0; i < F::NPAIR; ++i) {
                        const int k = t + i * F::LPF;
                        cf xa, xb;
                        F::r2c_pair(lds[f], k, HOIST ? ptw[i] : tb.w_n[tl + i * F::LPF], xa, xb);
                        xa = cscale(xa, hscale); xb = cscale(xb, hscale);
                        if constexpr (MODE == 0) {
                            o2[k] = xa;
                            o2[NC - k] = xb;
                        } else {
                            obase[k] = cnorm2(xa);
                            obase[NC - k] = cnorm2(xb);
                        }
                    }
                    if (t == 0) {
                        cf xm, unused;
                        F::r2c_pair(lds[f], NC / 2, mkc(0.0f, -1.0f), xm, unused);
                        xm = cscale(xm, hscale);
                        if constexpr (MODE == 0) o2[NC / 2] = xm;
                        else obase[NC / 2] = cnorm2(xm);
                    }
                }
            } else {
                // two-sided output, |X|^p with p != 2, or a dB epilogue: gather first, then finish in rolled loops
                cf xa[F::NPAIR], xb[F::NPAIR], xm, unused;
#pragma unroll
                for (int i = 0; i < F::NPAIR; ++i) {
                    F::r2c_pair(lds[f], t + i * F::LPF, HOIST ? ptw[i] : tb.w_n[tl + i * F::LPF], xa[i], xb[i]);
                    xa[i] = cscale(xa[i], hscale); xb[i] = cscale(xb[i], hscale);
                }
                F::r2c_pair(lds[f], NC / 2, mkc(0.0f, -1.0f), xm, unused);
                xm = cscale(xm, hscale);
                if constexpr (MODE == 0) {
                    if (live) {
#pragma unroll
                        for (int i = 0; i < F::NPAIR; ++i) {
                            const int k = t + i * F::LPF;
                            o2[k] = xa[i];
                            o2[NC - k] = xb[i];
                            if (k > 0) {            // mirror bins N-k = conj(X[k]), 0 < k < NC
                                o2[2 * NC - k] = mkc(xa[i].x, -xa[i].y);
                                o2[NC + k] = mkc(xb[i].x, -xb[i].y);
                            }
                        }
                        if (t == 0) {
                            o2[NC / 2] = xm;
                            o2[NC + NC / 2] = mkc(xm.x, -xm.y);
                        }
                    }
                } else {
                    wave_lds_fence();       // all Z reads of this frame are done: reuse its buffer as the |X|^2 row
#pragma unroll
                    for (int i = 0; i < F::NPAIR; ++i) {
                        const int k = t + i * F::LPF;
                        prow[k] = cnorm2(xa[i]);
                        prow[NC - k] = cnorm2(xb[i]);
                    }
                    if (t == 0) prow[NC / 2] = cnorm2(xm);
                    wave_lds_fence();
                    if (live) finish_power_row<NC, F::LPF>(prow, obase, t, nbins, ep.power, ep.db, ep.amin, ep.log10_ref);
                }
            }
        }
        wave_lds_fence();   // next iteration's first-pass writes must follow these reads
        st.mark(11);                                        // output row streamed out (store issue)
        unit = nxt_unit;
    }
}

// ---------------------------------------------------------------- software-pipelined variant
// One frame per wave (n_fft = 2048) with the plain epilogue (one-sided complex, or |X|^2).  The phase stamps of
// the kernel above (tools/stft_phase_timing.py) showed where a frame's ~13k cycles went: 40 % waiting for its
// own samples — the loads were issued right behind the previous frame's stores, and gfx950's single in-order
// vmcnt makes "my loads have landed" imply "all older stores are acknowledged" — and 22 % issuing those stores.
// Here the NEXT frame's samples are requested before the current frame's stores, every store is unconditional
// (clamped duplicate lanes instead of predication) so the compiler can wait with an exact vmcnt(#stores) while
// the stores drain behind the next frame's butterflies, and the window comes from LDS (lgkmcnt, not vmcnt).
// (A three-waves-per-SIMD form with the twiddles in LDS measured equal, 0.163 vs 0.158 ms, and was dropped: tools/ablation/.)
template <int NC, int E, int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 2)
stft_pipe_kernel(FrameGeom g, Tables tb, StftEpilogue ep) {
    using F = WaveFft<NC, E>;
    static_assert(F::G == 1 && radix_at(NC, 0) == E, "one frame per wave, single first-pass butterfly per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    const int t = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WAVE_SLOTS = ((F::PADDED + 1) / 2) * 2;
    cf* const lds = smem + w * WAVE_SLOTS;
    // window pairs, one 144-byte row per first-pass column: a lane's 16 values are 8 conflict-free ds_read_b128
    cf* const wlds = reinterpret_cast<cf*>(smem + WAVES * WAVE_SLOTS);
    constexpr int WROW = E + 2;
    for (int m = threadIdx.x; m < NC; m += WAVES * 64) wlds[(m & 63) * WROW + (m >> 6)] = window_pair(g, m);

    cf tw[F::NTW];
    cf ptw[F::NPAIR];
    F::load_twiddles(tw, tb.w_nc, t);
#pragma unroll
    for (int i = 0; i < F::NPAIR; ++i) ptw[i] = tb.w_n[t + i * F::LPF];

    const int T = (int)g.n_frames;
    const int total = (int)g.rows * T;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total ? begin + chunk : total;
    constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);
    constexpr int NST = ((LENF >> 2) + 63) / 64;          // 16-byte wave-stores per output row
    const float hscale = 0.5f * g.scale;                  // the R2C split returns 2·X

    // Frames are taken from a workgroup counter, not dealt out in fixed strides: where two waves share a SIMD the older
    // one wins the issue arbitration and would finish its share long before the other (melspec_stream.hpp).
    unsigned* const next_unit = reinterpret_cast<unsigned*>(wlds + 64 * WROW);
    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + WAVES);
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (t == 0) v = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    cf raw[E];
    bool pre = false;
    int unit = begin + w;
    if (unit < end) pre = prefetch_frame_raw_x<F>(raw, g, unit / T, unit % T, t);
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the loop is entered with nothing in flight
    __syncthreads();


    StftStamp st;
    while (unit < end) {
        const int nxt = grab();
        st.mark(0);
        const int urow = unit / T;
        const int uframe = unit - urow * T;
        cf v[1][E];
        cf* const ldsv[1] = {lds};
        if (pre) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4* wp = reinterpret_cast<const f4*>(wlds + t * WROW);
            f4 wv[E / 2];
#pragma unroll
            for (int i = 0; i < E / 2; ++i) wv[i] = wp[i];
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                v[0][2 * i] = cmul_elem(raw[2 * i], mkc(wv[i].x, wv[i].y));
                v[0][2 * i + 1] = cmul_elem(raw[2 * i + 1], mkc(wv[i].z, wv[i].w));
            }
        } else {
            load_frame<F, false>(v[0], g, nullptr, lds, urow, uframe, t);   // frames touching the padding
        }
        st.mark(8);
        // request the next frame as soon as this one's samples have left their registers: a whole frame of cover
        __builtin_amdgcn_sched_barrier(0);
        {
            pre = false;
            if (nxt < end) pre = prefetch_frame_raw_x<F>(raw, g, nxt / T, nxt % T, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        F::template run<1, StftStamp, true>(v, ldsv, tw, t, st, t);         // lower-half spectrum stays in registers
        st.mark(9);

        st.mark(1);                                         // next frame's loads issued

        const long long g0 = ((long long)urow * T + uframe) * LENF;
        const int a = (int)(g0 & 3);
        float* const stage = reinterpret_cast<float*>(lds) + a;      // LDS and global share their 16-byte phase
        {
            cf xa[F::NPAIR], xb[F::NPAIR], xm, unused;
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) {
                const int k = t + i * F::LPF;
                const cf zk = v[0][F::reg_of_spectrum(i)];            // Z[k] never left this lane
                const cf zm = (i == 0) ? F::r2c_partner(lds, k, zk) : lds[lds_pad(NC - k)];
                if constexpr (MODE != 0) {                            // xa[i] = (|X[k]|^2, |X[NC-k]|^2), no spectra formed
                    const cf pw = F::r2c_power_x2(zk, zm, ptw[i]);
                    xa[i] = cscale(pw, hscale * hscale);
                } else {
                    F::r2c_split_x2(zk, zm, ptw[i], xa[i], xb[i]);
                    xa[i] = cscale(xa[i], hscale); xb[i] = cscale(xb[i], hscale);
                }
            }
            F::r2c_pair(lds, NC / 2, mkc(0.0f, -1.0f), xm, unused);
            xm = cscale(xm, hscale);
            wave_lds_fence();                                         // every Z of this frame is in registers
            st.mark(7);                                               // R2C split done
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) {
                const int k = t + i * F::LPF;
                if constexpr (MODE == 0) {
                    reinterpret_cast<cf*>(stage)[k] = xa[i];
                    reinterpret_cast<cf*>(stage)[NC - k] = xb[i];
                } else {
                    stage[k] = spectral_row_value<MODE>(xa[i].x, ep);
                    stage[NC - k] = spectral_row_value<MODE>(xa[i].y, ep);
                }
            }
            if (t == 0) {
                if constexpr (MODE == 0) reinterpret_cast<cf*>(stage)[NC / 2] = xm;
                else stage[NC / 2] = spectral_row_value<MODE>(cnorm2(xm), ep);
            }
            wave_lds_fence();
        }
        st.mark(10);
        // the row leaves as 1 + NST + 1 unconditional stores: out-of-range lanes repeat a neighbour's element
        float* const gdst = ep.out + g0;
        const int npre = (4 - a) & 3;
        const int nchunks = (LENF - npre) >> 2;
        {
        {
            const int hmax = (npre > 1 ? npre : 1) - 1;
            const int hi = t < hmax ? t : hmax;
            gdst[hi] = stage[hi];
        }
        const float4* const s4 = reinterpret_cast<const float4*>(stage + npre);
        float4* const g4 = reinterpret_cast<float4*>(gdst + npre);
        static_assert(NST == 4 || NST == 8, "row store is written out for 4 or 8 wave-stores");
        const int last = nchunks - 1;
#define TAC_ROW_IDX(i) const int c##i = (t + 64 * i) < last ? (t + 64 * i) : last;
#define TAC_ROW_RD(i) const float4 b##i = s4[c##i];
#define TAC_ROW_WR(i) __builtin_nontemporal_store(__builtin_bit_cast(__attribute__((ext_vector_type(4))) float, b##i), \
            reinterpret_cast<__attribute__((ext_vector_type(4))) float*>(&g4[c##i]));
        TAC_ROW_IDX(0) TAC_ROW_IDX(1) TAC_ROW_IDX(2) TAC_ROW_IDX(3)
        TAC_ROW_RD(0) TAC_ROW_RD(1) TAC_ROW_RD(2) TAC_ROW_RD(3)
        if constexpr (NST == 8) {
            TAC_ROW_IDX(4) TAC_ROW_IDX(5) TAC_ROW_IDX(6) TAC_ROW_IDX(7)
            TAC_ROW_RD(4) TAC_ROW_RD(5) TAC_ROW_RD(6) TAC_ROW_RD(7)
            __builtin_amdgcn_sched_barrier(0);            // all LDS reads in flight before the first store issues
            TAC_ROW_WR(0) TAC_ROW_WR(1) TAC_ROW_WR(2) TAC_ROW_WR(3)
            TAC_ROW_WR(4) TAC_ROW_WR(5) TAC_ROW_WR(6) TAC_ROW_WR(7)
        } else {
            __builtin_amdgcn_sched_barrier(0);
            TAC_ROW_WR(0) TAC_ROW_WR(1) TAC_ROW_WR(2) TAC_ROW_WR(3)
        }
#undef TAC_ROW_IDX
#undef TAC_ROW_RD
#undef TAC_ROW_WR
        {
            const int r = LENF - npre - 4 * nchunks;
            const int rmax = (r > 1 ? r : 1) - 1;
            const int ti = LENF - 1 - (t < rmax ? t : rmax);
            gdst[ti] = stage[ti];
        }
        }
        wave_lds_fence();   // next iteration's first-pass writes must follow these reads
        st.mark(11);
        unit = nxt;
    }
}

constexpr int PIPE_WAVES = 8;      // one 8-wave workgroup per CU: both waves of every SIMD draw frames from the same counter

template <int NC, int E, int PMODE>
static int launch_pipe(const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, long long groups, hipStream_t stream) {
    using F = WaveFft<NC, E>;
    constexpr int WAVES = PIPE_WAVES;
    const size_t bytes = (size_t)WAVES * (((F::PADDED + 1) / 2) * 2) * sizeof(cf) + (size_t)64 * (E + 2) * sizeof(cf) + 16;
    long long blocks = (groups + WAVES - 1) / WAVES;
    const long long cap = (long long)device_cu_count() * (8 / WAVES);
    if (blocks > cap) blocks = cap;
    auto kern = stft_pipe_kernel<NC, E, PMODE, WAVES>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), bytes, stream, g, tb, ep);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// three waves per SIMD (stft_stream3.hpp); TAC_STFT_PIPE2=1 keeps the two-wave pipelined kernel above
template <int NC, int E, int PMODE>
static int launch_pipe3(const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, long long groups, hipStream_t stream) {
    static const bool two_wave = [] { const char* e = getenv("TAC_STFT_PIPE2"); return e && e[0] == '1'; }();
    // (rows shorter than a frame: the kernel's clamped whole-frame requests would have nothing to read)
    if (two_wave || g.length < 2 * NC) return launch_pipe<NC, E, PMODE>(g, tb, ep, groups, stream);
    // real rows: four waves per SIMD (0.134 -> 0.115 ms at cfg-2, inputs in the Infinity Cache); complex rows are bound by their
    // 658 MB of stores either way and measure best with three (profiles/r03/ab_stream3.txt)
    static const int waves_env = [] { const char* e = getenv("TAC_STFT_S3_WAVES"); return e ? atoi(e) : 0; }();
    const int waves = waves_env ? waves_env : (PMODE == 0 ? 12 : 16);
    // Row-store policy: nontemporal.  It keeps a cache-resident input resident (one re-read 164 MB batch: 0.150 / 0.113 ms
    // complex / power rows against 0.188 / 0.133 ms with plain stores, which allocate in the 256 MiB Infinity Cache and evict
    // the input), and with the input coming from HBM it measured equal or better than plain stores on every box of round 4
    // (same-process A/B, tools/r04/ab_inproc.py: 0.1854 vs 0.1944 ms, 0.1829 vs 0.19, equal on a third box) — round 3's
    // 6 % in favour of plain stores did not reproduce.  TAC_S3_STORES=nt|plain forces one form (A/B runs).
    static const int forced = [] { const char* e = getenv("TAC_S3_STORES"); return !e ? -1 : (e[0] == 'p' ? 1 : 0); }();
    const long long frames_total = g.rows * g.n_frames;
    const int plain = forced >= 0 ? forced : 0;
    auto go = [&](auto kern, int W, size_t bytes) {
        long long blocks = (groups + W - 1) / W;
        if (blocks > device_cu_count()) blocks = device_cu_count();
        if (blocks < 1) blocks = 1;
        const Stream3Launch lp{(frames_total + blocks - 1) / blocks, plain};
        TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(W * 64), bytes, stream, g, tb, ep, lp);
        TAC_HIP(hipGetLastError());
        return (int)TAC_OK;
    };
    {
        // Round 5: the samples through the LDS hop ring (stft_ring3.hpp) — every hop loaded ONCE per CU by a loader wave (LDS-DMA,
        // nontemporal) instead of four times by the frames that share it.  Same process against the forms below: complex rows
        // -6 ... -9 % (+ -3.8 % from the nontemporal policy), real rows -8.3 % (12 + 1 waves with the ring against 16 without),
        // bit-identical (profiles/r05/ab/batch14, batch17, batch19).  Conditions: hop = fft_length / 4 or / 8, whole hops of padding,
        // 16-byte aligned hops, 31-bit hop ids.
        static const bool off = [] { const char* e = getenv("TAC_S3_RING"); return e && e[0] == '0'; }();
        constexpr int TWv = TAC_S3_RING_TW;
        auto ring = [&](auto hpf_tag) {
            constexpr int HPF = decltype(hpf_tag)::value;
            using RC = Ring3Cfg<NC, E, PMODE, TWv, HPF>;
            long long blocks = (groups + TWv - 1) / TWv;
            if (blocks > device_cu_count()) blocks = device_cu_count();
            if (blocks < 1) blocks = 1;
            const Stream3Launch lp{(frames_total + blocks - 1) / blocks, plain};
            auto kern = stft_ring3_kernel<NC, E, PMODE, TWv, HPF>;
            set_last_route("stft_ring3_kernel<%d, %d, %d, %d, %d>", NC, E, PMODE, TWv, HPF);
            TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), RC::BYTES));
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3((TWv + 1) * 64), RC::BYTES, stream, g, tb, ep, lp);
            TAC_HIP(hipGetLastError());
            return (int)TAC_OK;
        };
        const int hpf = g.hop > 0 && (2 * NC) % g.hop == 0 ? 2 * NC / g.hop : 0;
        if (!off && !waves_env && (hpf == 4 || hpf == 8) && g.vec4_ok && (g.center_pad % g.hop) == 0 &&
            g.rows * (g.n_frames + 8) < 0x7fffffffLL && (reinterpret_cast<uintptr_t>(ep.out) & 15u) == 0) {
            if (hpf == 4) return ring(std::integral_constant<int, 4>{});
            return ring(std::integral_constant<int, 8>{});      // (hop = fft_length / 2: thirteen 4 KB hops do not fit beside twelve areas)
        }
    }
    set_last_route("stft_stream3_kernel<%d, %d, %d, %d>", NC, E, PMODE, waves == 12 ? 12 : 16);
    if (waves == 12) return go(stft_stream3_kernel<NC, E, PMODE, 12>, 12, stft_stream3_lds_bytes<NC, E, 12>());
    return go(stft_stream3_kernel<NC, E, PMODE, 16>, 16, stft_stream3_lds_bytes<NC, E, 16>());
}

template <int NC, int E, int MODE>
static int launch_stft(const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, hipStream_t stream) {
    using F = WaveFft<NC, E>;
    // frames in flight per wave (generic kernel): two fit since the packed-math core (223 registers, no scratch) and
    // measure 3 % faster without the pipelining; one is what ships
    constexpr int NF = 1;
    constexpr bool HOIST = (E <= 16);
    const long long groups = g.rows * ((g.n_frames + NF * F::G - 1) / (NF * F::G));     // wave-iterations
    if (groups >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    if constexpr (F::G == 1 && E == 16) {
        // pipelined kernel: one-sided complex rows, or |X| / |X|^2 rows with or without the dB epilogue
        int pmode = -1;
        if (ep.onesided) {
            if (MODE == 0) pmode = 0;
            else if (ep.power == 2.0f) pmode = ep.db ? 3 : 1;
            else if (ep.power == 1.0f) pmode = ep.db ? 4 : 2;
        }
        switch (pmode) {
            case 0: return launch_pipe3<NC, E, 0>(g, tb, ep, groups, stream);
            case 1: return launch_pipe3<NC, E, 1>(g, tb, ep, groups, stream);
            case 2: return launch_pipe3<NC, E, 2>(g, tb, ep, groups, stream);
            case 3: return launch_pipe3<NC, E, 3>(g, tb, ep, groups, stream);
            case 4: return launch_pipe3<NC, E, 4>(g, tb, ep, groups, stream);
            default: break;
        }
    }
    // 256-register waves: eight waves fill a CU.  One 8-wave workgroup per CU (both waves of a SIMD draw units from the
    // same counter) where its frame buffers fit the LDS and there is that much work, 4-wave workgroups otherwise.
    constexpr size_t wave_bytes = (size_t)(((NF * F::G * F::PADDED + 1) / 2) * 2) * sizeof(cf);
    constexpr bool WIDE_FITS = 2 * STFT_WAVES * wave_bytes + 16 <= 160 * 1024;
    const bool wide = WIDE_FITS && groups >= 2LL * STFT_WAVES * device_cu_count();
    const int waves = wide ? 2 * STFT_WAVES : STFT_WAVES;
    const size_t lds_bytes = (size_t)waves * wave_bytes + 16;
    int per_cu = wide ? 1 : (int)(160 * 1024 / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    long long max_blocks = (long long)device_cu_count() * per_cu;
    long long want = (groups + waves - 1) / waves;
    long long blocks = want < max_blocks ? want : max_blocks;
    if (blocks < 1) blocks = 1;
    auto launch = [&](auto kern) {
        if (lds_bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds_bytes));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(waves * 64), lds_bytes, stream, g, tb, ep);
        TAC_HIP(hipGetLastError());
        return (int)TAC_OK;
    };
    if constexpr (WIDE_FITS) {
        if (wide) return launch(stft_kernel<NC, E, MODE, NF, HOIST, 2 * STFT_WAVES>);
    }
    return launch(stft_kernel<NC, E, MODE, NF, HOIST, STFT_WAVES>);
}

int try_launch_n4096(const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream);
int try_launch_small(int n_fft, const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, int mode, hipStream_t stream);
int try_launch_n400(const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream);   // stft_n400.hip
int launch_stft_big(int n_fft, const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream);   // stft_big.hip
bool stft_smooth_covers(int n_fft);                                                                           // stft_smooth.hip
int launch_stft_smooth(int n_fft, const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream);

static inline bool is_big_fft(int n_fft) { return n_fft == 8192 || n_fft == 16384 || n_fft == 32768; }

template <int MODE>
static int dispatch_stft(int n_fft, const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, hipStream_t s) {
    switch (n_fft) {
        case 32: return launch_stft<16, 16, MODE>(g, tb, ep, s);
        case 64: return launch_stft<32, 16, MODE>(g, tb, ep, s);
        case 128: return launch_stft<64, 16, MODE>(g, tb, ep, s);
        case 256: {
            const int rc = try_launch_small(n_fft, g, tb, ep, MODE, s);     // stft_small3.hpp: plain epilogues, rows >= one frame
            if (rc != TAC_E_UNSUPPORTED) return rc;
            return launch_stft<128, 16, MODE>(g, tb, ep, s);
        }
        case 512:
        case 1024: {
            const int rc = try_launch_small(n_fft, g, tb, ep, MODE, s);     // stft_small.hip: plain epilogues
            if (rc != TAC_E_UNSUPPORTED) return rc;
            return n_fft == 512 ? launch_stft<256, 16, MODE>(g, tb, ep, s) : launch_stft<512, 16, MODE>(g, tb, ep, s);
        }
        case 2048: return launch_stft<1024, 16, MODE>(g, tb, ep, s);
        case 4096: {
            const int rc = try_launch_n4096(g, ep, MODE, s);          // stft_n4096.hip: plain epilogues, aligned frames
            if (rc != TAC_E_UNSUPPORTED) return rc;
            return launch_stft<2048, 32, MODE>(g, tb, ep, s);
        }
        case 400: return try_launch_n400(g, ep, MODE, s);          // 200 = 8 x 25 mixed radix; plain one-sided epilogues
        case 8192:
        case 16384:
        case 32768: return launch_stft_big(n_fft, g, ep, MODE, s);      // four-step transform, one frame per workgroup
        default:                                                    // even lengths with a 7-smooth half: generic Stockham passes
            return stft_smooth_covers(n_fft) ? launch_stft_smooth(n_fft, g, ep, MODE, s) : TAC_E_UNSUPPORTED;
    }
}

}  // namespace tac

extern "C" {

int tac_stft_f32(const float* wave, const float* window, const tac_stft_desc* d, float* out, void* stream) {
    using namespace tac;
    if (!out) return TAC_E_INVALID;
    FrameGeom g;
    int64_t T = 0;
    int rc = make_geometry(wave, window, d, &g, &T, d && (is_big_fft(d->n_fft) || stft_smooth_covers(d->n_fft)));
    if (rc != TAC_OK) return rc;
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    StftEpilogue ep{out, d->onesided ? 1 : 0, 0, 1.0f, 0, 0.0f, 0.0f};
    return dispatch_stft<0>(d->n_fft, g, tb, ep, (hipStream_t)stream);
}

int tac_spectrogram_f32(const float* wave, const float* window, const tac_stft_desc* d, float power, int db,
                        float db_ref, float db_amin, float* out, void* stream) {
    using namespace tac;
    if (!out) return TAC_E_INVALID;
    FrameGeom g;
    int64_t T = 0;
    int rc = make_geometry(wave, window, d, &g, &T, d && (is_big_fft(d->n_fft) || stft_smooth_covers(d->n_fft)));
    if (rc != TAC_OK) return rc;
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    StftEpilogue ep{out, d->onesided ? 1 : 0, 1, power, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f};
    return dispatch_stft<1>(d->n_fft, g, tb, ep, (hipStream_t)stream);
}

}  // extern "C"
