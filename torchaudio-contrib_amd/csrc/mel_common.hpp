// mel_common.hpp — pieces shared by the two fused Melspectrogram kernels (MFMA and band-sparse VALU
// contraction): the frame-buffer geometry and phase A (FFT of a tile's frames, in-place power rows).
#pragma once
#include "host_common.hpp"

namespace tac {

constexpr int MEL_STEP_BUDGET = 384;         // K-steps per workgroup held in registers during MFMA phase B
constexpr int MEL_CHUNK = 8;                 // K-step granularity of the per-wave capacity

// A workgroup owns tiles of TILE consecutive frames of one row; LDS holds one buffer per frame which is first the
// FFT exchange area and then, in place, the frame's |X|^p row (row stride PROW floats == 2 mod 32).
template <int NC, int E, int TILE, int BUDGET = MEL_STEP_BUDGET>
struct MelCfg {
    using F = WaveFft<NC, E>;
    static constexpr int TILE_FRAMES = TILE;
    static constexpr int GT = (TILE / F::G) >= 1 ? (TILE / F::G) : 1;      // lane-groups (waves' worth) per tile
    static constexpr int WAVES = (TILE == 8) ? 4 : (GT >= 8 ? 8 : 4);
    static constexpr int GPW = (GT / WAVES) >= 1 ? (GT / WAVES) : 1;       // groups (sequential FFT rounds) per wave
    static constexpr int NBUF = (WAVES * GPW * F::G) > TILE ? (WAVES * GPW * F::G) : TILE;
    static constexpr int PROW = 2 * F::PADDED;                             // floats between consecutive P rows
    static constexpr int CAP = BUDGET / WAVES;                             // K-steps per wave (register-resident)
    static constexpr int NCHUNK = CAP / MEL_CHUNK;
    static constexpr int SLOT = TILE * 16;                                 // floats per partial slot
    static_assert(PROW >= NC + 1 + 7, "P row must hold F bins + contraction overrun");
    static_assert(CAP % MEL_CHUNK == 0, "step capacity must be whole chunks");
};

// Lane-invariant FFT constants of a fused kernel (registers for the kernel's lifetime).
// HOISTW: window register-resident too.  Kernels with spare registers (HOISTW) also keep all E/2 R2C twiddles
// instead of the factored form (one register x compile-time constants), saving its extra complex multiplies.
template <class F, bool HOISTW = false, bool FORCE_FACT = false>
struct MelFftConsts {
    static constexpr bool FACT = (!HOISTW || FORCE_FACT) && (F::E == 16) && (F::LPF * 32 == 2 * F::NC);   // W_N^{i*LPF} == W_32^i
    static constexpr bool HOIST_WINDOW = HOISTW;
    cf tw[F::NTW];
    cf ptw[FACT ? 1 : F::NPAIR];
    cf win[HOISTW ? F::E : 1];            // window pairs of this lane's elements (kernels with spare registers)
    __device__ __forceinline__ void load(const Tables& tb, const FrameGeom& g, int t) {
        if constexpr (HOISTW) load_window_regs<F>(win, g, t);
        F::load_twiddles(tw, tb.w_nc, t);
        if constexpr (FACT) {
            ptw[0] = tb.w_n[t];
        } else {
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) ptw[i] = tb.w_n[t + i * F::LPF];
        }
    }
};

// Phase A of one tile: every wave FFTs its frames and overwrites each frame buffer with the power row.
// (Advancing two of a wave's frames together fits the registers since the packed-math core but measures 3-6 % slower.)
// PIPE (G == 1 kernels): every frame's raw samples are requested one frame ahead — right after the previous frame's
// butterflies, before its R2C/power epilogue — so the HBM/L2 round trip never opens a frame.  raw/pre_ok carry the
// request across calls; (next_row, next_f0) name the tile whose first frame of this wave follows this tile's last
// (next_f0 < 0: none).  The frame loop is fully unrolled so the compiler can count the stores between a request
// and its use (gfx950's single in-order vmcnt: an uncounted wait would also wait for those stores' acknowledgements).
template <class C, bool POW2, bool HOISTW = false, class ST = NoStamp, bool PIPE = false, class K = void>
__device__ __forceinline__ void mel_phase_a(const FrameGeom& g, cf* bufs, const K& k,
                                            int w, int sub, int t, int row, long long f0,
                                            cf* pre_raw = nullptr, bool* pre_ok_p = nullptr, ST* stp = nullptr,
                                            int next_row = 0, long long next_f0 = -1) {
    ST st_local;
    ST& st = stp ? *stp : st_local;
    bool pre_ok = pre_ok_p ? *pre_ok_p : false;
    using F = typename C::F;
    constexpr int NC = F::NC, E = F::E, NBINS = NC + 1, TILE = C::TILE_FRAMES;
    constexpr bool FACT = K::FACT;
    static_assert(K::HOIST_WINDOW == HOISTW, "window hoisting of the constants and the caller disagree");
    constexpr int NF = 1;                                 // frames in flight per wave (WaveFft::run's batch)
    const cf* tw = k.tw;
    const cf* ptw = k.ptw;
    const bool wave_has_frames = (w * C::GPW * F::G) < TILE;
    if (wave_has_frames) {
#pragma unroll PIPE ? C::GPW : 1
        for (int rep = 0; rep < C::GPW; rep += NF) {
            // the older wave of a SIMD wins VALU arbitration, so after the first frame the younger one is behind:
            // it gets the priority for the tile's second frame and both reach the barrier together (-1.5 %)
            if (PIPE && C::WAVES == 8 && rep + NF >= C::GPW) { if (w >= 4) __builtin_amdgcn_s_setprio(2); }
            cf* lds[NF];
            int fi[NF];
            cf v[NF][E];
            int tl = t;
            asm volatile("" : "+v"(tl));      // launder: window loads stay inside the loop (register budget)
            cf winl[HOISTW ? 1 : F::E];
            if constexpr (!HOISTW) load_window_regs<F>(winl, g, PIPE ? t : tl);
            const cf* win = HOISTW ? k.win : winl;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                fi[f] = ((w * C::GPW + rep + f) * F::G) + sub;              // frame index within the tile
                lds[f] = bufs + fi[f] * F::PADDED;
                if (PIPE && pre_ok) {
                    apply_window<F>(v[f], pre_raw, win);
                } else {
                    load_frame<F, true>(v[f], g, win, lds[f], row, (fi[f] < TILE) ? f0 + fi[f] : g.n_frames, t);
                }
            }
            st.mark(8);
            F::template run<NF, ST, true>(v, lds, tw, t, st, t);   // lower-half spectrum stays in registers
            st.mark(9);
            if constexpr (PIPE) {
                static_assert(NF == 1, "pipelined phase A advances one frame group per wave-round");
                __builtin_amdgcn_sched_barrier(0);
                const bool same_tile = rep + 1 < C::GPW;
                const int nrow = same_tile ? row : next_row;
                const long long nfi = (long long)(w * C::GPW + (same_tile ? rep + 1 : 0)) * F::G + sub;
                const long long nframe = (same_tile ? f0 : next_f0) + nfi;
                pre_ok = false;
                if ((same_tile || next_f0 >= 0) && nfi < TILE)
                    pre_ok = prefetch_frame_raw_x<F>(pre_raw, g, nrow, nframe, t);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                // gather every partner Z[NC-k] this lane needs BEFORE anything is overwritten (in-place row); the
                // Z[k] themselves never left the lane (WaveFft::run<..., HALF>)
                cf zk[F::NPAIR], zm[F::NPAIR];
#pragma unroll
                for (int i = 0; i < F::NPAIR; ++i) {
                    const int kk = t + i * F::LPF;
                    zk[i] = v[f][F::reg_of_spectrum(i)];
                    zm[i] = (i == 0) ? F::r2c_partner(lds[f], kk, zk[i]) : lds[f][lds_pad(NC - kk)];
                }
                const cf zmid = lds[f][lds_pad(NC / 2)];
                const float pfac = 0.25f * g.scale * g.scale;              // |2X|^2 -> |scale·X|^2
                wave_lds_fence();
                float* prow = reinterpret_cast<float*>(lds[f]);
#pragma unroll
                for (int i = 0; i < F::NPAIR; ++i) {
                    const int kk = t + i * F::LPF;
                    cf xa, xb;
                    // xa, xb = 2·X: the halving and the `normalized` scale are one factor applied to the power
                    if constexpr (FACT) F::r2c_split_factored_x2(zk[i], zm[i], ptw[0], i, xa, xb);
                    else F::r2c_split_x2(zk[i], zm[i], ptw[FACT ? 0 : i], xa, xb);
                    const float pa = cnorm2(xa) * pfac, pb = cnorm2(xb) * pfac;
                    prow[kk] = POW2 ? pa : sqrtf(pa);
                    prow[NC - kk] = POW2 ? pb : sqrtf(pb);
                }
                if (t == 0) {
                    const cf xm = mkc(zmid.x * g.scale, -zmid.y * g.scale);    // X[NC/2] = conj(Z[NC/2])
                    const float pm = xm.x * xm.x + xm.y * xm.y;
                    prow[NC / 2] = POW2 ? pm : sqrtf(pm);
                }
                for (int c = t; c < 7; c += F::LPF) prow[NBINS + c] = 0.0f;           // contraction overrun columns
            }
            st.mark(10);
        }
    }
    if (PIPE && C::WAVES == 8) __builtin_amdgcn_s_setprio(0);
    if (pre_ok_p) *pre_ok_p = pre_ok;
}

}  // namespace tac
