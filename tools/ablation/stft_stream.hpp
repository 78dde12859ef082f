// stft_stream.hpp — one-sided complex STFT / |X|^p (/dB) rows for fft_length 2048 in the streaming form of
// melspec_stream.hpp: every wave carries two frames half a frame apart through  s0 (window, pass 0, exchange) ·
// s12 (pass 1, in-register exchange, pass 2, R2C partner reads) · s3 (R2C split, epilogue, row stores), rotating
//     A.s0 | B.s12 | A.s12 | B.s3 | A.s3 | B.s0
// so that each stage's LDS round trip was issued one other-frame stage earlier, with one shared exchange area per wave,
// no workgroup barrier, and the next frame's samples requested into the frame's own (dead) registers right after its
// rows are stored.  Rows leave straight from registers: for a pair index p the 64 lanes hold 64 consecutive bins
// (t + 64p ascending, NC - t - 64p descending), i.e. every store instruction is one contiguous 256-byte (512-byte for
// complex rows) run — nontemporal, the rows are written once and never re-read.
#pragma once
#include "../../torchaudio-contrib_amd/csrc/host_common.hpp"

#include <type_traits>

namespace tac {

constexpr int SS_WAVES = 8;

// MODE: 0 complex rows, 1 |X|^2, 2 |X|, 3 |X|^2 in dB, 4 |X| in dB (spectral_row_value, host_common.hpp)
template <int NC, int E, int MODE>
__global__ void __launch_bounds__(SS_WAVES * 64, 2)
stft_stream_kernel(FrameGeom g, Tables tb, StftEpilogue ep, long long total) {
    using F = WaveFft<NC, E>;
    static_assert(F::G == 1, "one frame per wave-pass");
    constexpr int NBINS = NC + 1, SLOTS = 2 * SS_WAVES;
    constexpr int XA_BYTES = (F::PADDED * sizeof(cf) + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cf* xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);

    const long long chunk = (total + gridDim.x - 1) / gridDim.x;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < total ? begin + chunk : total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    const unsigned T = (unsigned)g.n_frames;

    const int t = lane;
    cf tw[F::NTW];
    F::load_twiddles(tw, tb.w_nc, t);
    const cf w0 = tb.w_n[t];
    cf win[E];
    load_window_regs<F>(win, g, t);
    const float half = 0.5f * g.scale;                                 // 2X -> scale * X once, in the window
#pragma unroll
    for (int e = 0; e < E; ++e) win[e] = cscale(win[e], half);

    unsigned* const next_frame = reinterpret_cast<unsigned*>(smem_raw + (size_t)SS_WAVES * XA_BYTES);
    if (threadIdx.x == 0) *next_frame = SLOTS;                          // frames beyond the first SLOTS come from this counter
    __syncthreads();
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    cf vA[E], vB[E];
    cf zmA[F::NPAIR], zmB[F::NPAIR], zmidA = mkc(0.f, 0.f), zmidB = mkc(0.f, 0.f);

    // unconditional, clamped sample request (see melspec_stream.hpp: the loop stays branch-free so waits are counted)
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
        cs = cs + F::N <= g.length ? cs : g.length - F::N;
        const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)row * g.row_stride + cs);
#pragma unroll
        for (int q = 0; q < E; ++q) raw[q] = src[t + q * F::LPF];
    };
    auto s0 = [&](cf (&v)[E], int mode, int row, long long fr) {
        if (mode == 1) {
            apply_window<F>(v, v, win);
        } else {
            load_frame<F, true>(v, g, win, xa, row, fr, t);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = cscale(v[e], half);
        }
        F::template pass_butterflies<0>(v);
        wave_lds_fence();
        F::template pass_write<0, true>(v, xa, t, t);
        wave_lds_fence();
        F::template pass_readback<1>(v, xa, t);
    };
    auto s12 = [&](cf (&v)[E], cf (&zm)[F::NPAIR], cf& zmid) {
        F::template pass_twiddle<1>(v, tw);
        F::template pass_butterflies<1>(v);
        F::exchange_1_2_in_registers(v);
        F::template pass_twiddle<2>(v, tw);
        F::template pass_butterflies<2>(v);
        wave_lds_fence();
        F::template pass_write<2, true>(v, xa, t, t);
        wave_lds_fence();
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            const int kk = t + p * F::LPF;
            zm[p] = (p == 0) ? F::r2c_partner(xa, kk, v[F::reg_of_spectrum(0)]) : xa[lds_pad(NC - kk)];
        }
        zmid = xa[lds_pad(NC / 2)];
    };
    // s3: R2C split + epilogue into registers; the rows are stored AFTER the next frame's samples have been requested
    // (gfx950 completes vector-memory operations in order: loads requested behind the stores would also wait for the
    // stores' acknowledgements)
    using RowT = typename std::conditional<MODE == 0, cf, float>::type;
    auto s3 = [&](cf (&v)[E], cf (&zm)[F::NPAIR], cf zmid, RowT (&lo)[F::NPAIR], RowT (&hi)[F::NPAIR], RowT& mid) {
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            cf xk, xm;
            F::r2c_split_factored_x2(v[F::reg_of_spectrum(p)], zm[p], w0, p, xk, xm);
            if constexpr (MODE == 0) {
                lo[p] = xk;
                hi[p] = xm;
            } else {
                lo[p] = spectral_row_value<MODE>(cnorm2(xk), ep);
                hi[p] = spectral_row_value<MODE>(cnorm2(xm), ep);
            }
        }
        if constexpr (MODE == 0) mid = mkc(2.0f * zmid.x, -2.0f * zmid.y);
        else mid = spectral_row_value<MODE>(4.0f * cnorm2(zmid), ep);
    };
    auto store_rows = [&](const RowT (&lo)[F::NPAIR], const RowT (&hi)[F::NPAIR], RowT mid, int i) {
        i = i < nloc ? i : nloc - 1;
        RowT* orow = reinterpret_cast<RowT*>(ep.out) + (begin + i) * NBINS;
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            const int kk = t + p * F::LPF;
            __builtin_nontemporal_store(lo[p], orow + kk);
            __builtin_nontemporal_store(hi[p], orow + (NC - kk));
        }
        if (t == 0) __builtin_nontemporal_store(mid, orow + NC / 2);
    };

    if (nloc > 0) {
        RowT rlo[F::NPAIR], rhi[F::NPAIR], rmid;
        int modeA, rowA, modeB, rowB;
        long long frA, frB;
        int iA = 2 * w, iB = 2 * w + 1;
        request(vB, iB, modeB, rowB, frB);
        request(vA, iA, modeA, rowA, frA);
        s0(vB, modeB, rowB, frB);
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): the loop is entered with nothing in flight
#pragma unroll 1
        while (iA < nloc || iB < nloc) {
            s0(vA, modeA, rowA, frA);
            __builtin_amdgcn_sched_barrier(0);
            s12(vB, zmB, zmidB);
            __builtin_amdgcn_sched_barrier(0);
            s12(vA, zmA, zmidA);
            __builtin_amdgcn_sched_barrier(0);
            s3(vB, zmB, zmidB, rlo, rhi, rmid);
            const int nB = grab();
            request(vB, nB, modeB, rowB, frB);
            store_rows(rlo, rhi, rmid, iB);
            iB = nB;
            __builtin_amdgcn_sched_barrier(0);
            s3(vA, zmA, zmidA, rlo, rhi, rmid);
            const int nA = grab();
            request(vA, nA, modeA, rowA, frA);
            store_rows(rlo, rhi, rmid, iA);
            iA = nA;
            __builtin_amdgcn_sched_barrier(0);
            s0(vB, modeB, rowB, frB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NC, int E, int MODE>
static int launch_stft_stream(const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, hipStream_t stream) {
    using F = WaveFft<NC, E>;
    const long long total = g.rows * g.n_frames;
    if (total >= 0x7fffffffLL || g.length < 2 * NC) return TAC_E_UNSUPPORTED;
    const size_t lds_bytes = (size_t)SS_WAVES * ((F::PADDED * sizeof(cf) + 15) & ~(size_t)15) + 16;
    long long blocks = (total + 2 * SS_WAVES - 1) / (2 * SS_WAVES);
    if (blocks > device_cu_count()) blocks = device_cu_count();
    if (blocks < 1) blocks = 1;
    auto kern = stft_stream_kernel<NC, E, MODE>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SS_WAVES * 64), lds_bytes, stream, g, tb, ep, total);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // namespace tac
