// backward_ring3_multi.hpp — melspec_backward_ring3_kernel (backward_ring3.hpp) for fft_length 512 / 1024, where a wave
// carries G = 4 / 2 frames side by side (LPF = 16 / 32 lanes each): twelve waves per CU, every lane group walks its OWN
// segment of consecutive frames with its overlap-add ring in 16 - H register pairs per lane (H = hop / (fft_length / 16), a
// template parameter), the filterbank adjoint (FUSE) formed per bin pair inside the kernel.  Replaces
// spectrogram_backward_ola_multi_kernel (+ fb_adjoint_kernel) for hop = fft_length / 8, / 4, / 2.
// Frame, row and the segment flags are per lane here; a group whose segment is shorter than its neighbours' recomputes its
// last frame and stores nothing until the longest is done.
#pragma once
#include "backward_ring3.hpp"

namespace tac {

template <int NC>
__host__ __device__ inline size_t ring3_multi_lds_bytes(int mel_stride) {
    using F = WaveFft<NC, 16>;
    constexpr int WAVE_SLOTS = ((F::G * F::PADDED + 1) / 2) * 2;
    return (size_t)BR_WAVES * WAVE_SLOTS * sizeof(cf) + (size_t)F::LPF * 18 * sizeof(cf) + ST_TW_BYTES +
           (size_t)4 * F::LPF * 16 + (size_t)(NC + 1) * sizeof(AdjEntry) + (size_t)BR_WAVES * F::G * mel_stride * sizeof(float) + 16;
}

template <int NC, bool POW2, int H, bool FUSE>
__global__ void __launch_bounds__(BR_WAVES * 64, 3)
melspec_backward_ring3_multi_kernel(FrameGeom g, Tables tb, const float* __restrict__ gmel, float power,
                                    float* __restrict__ gpad, float* __restrict__ edge, OlaPlan plan, OlaFuse fz) {
    constexpr int E = 16, N = 2 * NC, NBINS = NC + 1, R = 16 - H, WAVES = BR_WAVES;
    using F = WaveFft<NC, E>;
    constexpr int LPF = F::LPF, G = F::G, NPASS = F::NPASS, R1 = radix_at(NC, 1);
    static_assert(G >= 2 && (NPASS == 2 || NPASS == 3) && R1 == 16 && LPF % 16 == 0, "fft_length 512 / 1024");
    static_assert(H >= 1 && H < 16, "hop = (fft_length / 16) H < fft_length");
    constexpr int MELQ = 128 / LPF;                         // mel-gradient values per lane (n_mels <= 128)
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* const smem = reinterpret_cast<cf*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / LPF, t = lane % LPF;
    constexpr int WAVE_SLOTS = ((G * F::PADDED + 1) / 2) * 2;
    cf* const lds = smem + w * WAVE_SLOTS + sub * F::PADDED;
    constexpr int WROW = E + 2;
    const float half = 0.5f * g.scale;      // forward: the R2C split returns 2X; inverse: the common 1/2 of the C2R operands
    cf* const wlds = smem + WAVES * WAVE_SLOTS;
    for (int m = threadIdx.x; m < NC; m += WAVES * 64) wlds[(m % LPF) * WROW + (m / LPF)] = cscale(window_pair(g, m), half);
    float* const twlds = reinterpret_cast<float*>(wlds + LPF * WROW);
    if (threadIdx.x < 16 * 16) {
        const int js = threadIdx.x >> 4, q = threadIdx.x & 15;
        const cf wv = q ? tb.w_nc[js * q * (NC / 256)] : mkc(1.0f, 0.0f);
        twlds[js * ST_TW_STRIDE + 2 * (q ? q - 1 : 15)] = wv.x;
        twlds[js * ST_TW_STRIDE + 2 * (q ? q - 1 : 15) + 1] = wv.y;
    }
    cf* const ptwl = reinterpret_cast<cf*>(twlds + ST_TW_BYTES / 4);
    for (int idx = threadIdx.x; idx < LPF * F::NPAIR; idx += WAVES * 64) {
        const int tt = idx / F::NPAIR, p = idx - tt * F::NPAIR;
        ptwl[((p >> 1) * LPF + tt) * 2 + (p & 1)] = tb.w_n[tt + p * LPF];
    }
    AdjEntry* const adj_lds = reinterpret_cast<AdjEntry*>(ptwl + 2 * 4 * LPF);
    if constexpr (FUSE)
        for (int k = threadIdx.x; k < NBINS; k += WAVES * 64) adj_lds[k] = fz.adj[k];
    float* const grow = reinterpret_cast<float*>(adj_lds + NBINS) + (w * G + sub) * fz.mel_stride;
    cf tw2 = mkc(1.0f, 0.0f);
    if constexpr (NPASS == 3) {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
        tw2 = all[twiddles_before(NC, E, 2)];
    }
    __syncthreads();

    const int T = (int)g.n_frames, hop = g.hop, S = plan.seg_frames, spr = plan.segs_per_row;
    const long long nseg_total = g.rows * (long long)spr;
    const long long ngroups = (nseg_total + G - 1) / G;

    // passes after the first on the outputs of pass 0 (in v); HALF: the lower half of the result stays in registers
    auto passes_after_first = [&](cf (&v)[E], auto half_tag) {
        constexpr bool HALF = decltype(half_tag)::value;
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
        F::template pass_write<0, true>(v, lds, t, t);
        wave_lds_fence();
        F::template pass_readback<1>(v, lds, t);
        F::template pass_twiddle<1, true>(v, tw1);
        F::template pass_butterflies<1>(v);
        wave_lds_fence();
        F::template pass_write<1, HALF>(v, lds, t, t);
        wave_lds_fence();
        if constexpr (NPASS == 3) {
            F::template pass_readback<2>(v, lds, t);
            F::template pass_twiddle<2, true>(v, &tw2);
            F::template pass_butterflies<2>(v);
            wave_lds_fence();
            F::template pass_write<2, HALF>(v, lds, t, t);
            wave_lds_fence();
        }
    };

    cf v[E];
    bool fast = false;
    // samples (+ mel-gradient row) of (row r, frame fr) — per lane group —, unconditionally, from a clamped address
    auto request = [&](int r, int fr) {
        const long long start = (long long)fr * hop - g.center_pad;
        const bool ok = g.vec2_ok && start >= 0 && start + N <= g.length;
        fast = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        long long cs = start < 0 ? 0 : start;
        cs = cs + N <= g.length ? cs : g.length - N;
        const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)r * g.row_stride + cs);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            if (q < H) v[q] = __builtin_nontemporal_load(src + t + q * LPF);      // the frame's oldest hop: no later frame reads it
            else v[q] = src[t + q * LPF];
        }
    };

    for (long long gi = (long long)blockIdx.x * WAVES + w; gi < ngroups; gi += (long long)gridDim.x * WAVES) {
        const long long seg = gi * G + sub;
        const bool seg_ok = seg < nseg_total;
        const long long segc = seg_ok ? seg : nseg_total - 1;
        const int row = (int)(segc / spr), sidx = (int)(segc - (long long)row * spr);
        const int f0 = sidx * S, f1 = f0 + S < T ? f0 + S : T;
        const int len_lane = seg_ok ? f1 - f0 : 0;
        int len = 0;                                                       // the longest of the wave's segments
#pragma unroll
        for (int s2 = 0; s2 < G; ++s2) {
            const int l2 = __builtin_amdgcn_readlane(len_lane, s2 * LPF);
            len = l2 > len ? l2 : len;
        }
        const bool row_end = (f1 == T);
        float* const tail0 = edge + ((long long)row * (spr - 1) + sidx) * (N - hop) - hop;
        cf ring[R];
#pragma unroll
        for (int r = 0; r < R; ++r) ring[r] = mkc(0.0f, 0.0f);
        request(row, f0);
        for (int i = 0; i < len; ++i) {
            const bool live = i < len_lane, last = (i + 1 == len_lane);
            const int f = live ? f0 + i : f0;
            float gq[MELQ];
            if constexpr (FUSE) {                                       // the frame's mel-gradient row, likewise
                const float* gn = gmel + ((long long)row * T + f) * fz.n_mels;
#pragma unroll
                for (int q = 0; q < MELQ; ++q) {
                    const int b = t + LPF * q;
                    gq[q] = br3_load_once(gn + (b < fz.n_mels ? b : fz.n_mels - 1));
                }
            }
            float gk[F::NPAIR], gm[F::NPAIR], gmid_reg = 0.0f;
            if constexpr (!FUSE) {                                      // the gradient values of this lane's pairs travel behind the forward transform
                const float* gn = gmel + ((long long)row * T + f) * NBINS;
#pragma unroll
                for (int p = 0; p < F::NPAIR; ++p) {
                    gk[p] = br3_load_once(gn + t + p * LPF);
                    gm[p] = br3_load_once(gn + NC - (t + p * LPF));
                }
                gmid_reg = br3_load_once(gn + NC / 2);
            }
            // ---- forward transform of the group's frames
            if (!fast) {
                int tz;
                asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
                load_frame<F, false, true, true>(v, g, nullptr, lds, row, f, tz, FetchF32{g.wave});
            }
            {
                const f4* wp = reinterpret_cast<const f4*>(wlds + t * WROW);
                cf win[E];
#pragma unroll
                for (int u = 0; u < E / 2; ++u) {
                    const f4 x = wp[u];
                    win[2 * u] = mkc(x.x, x.y);
                    win[2 * u + 1] = mkc(x.z, x.w);
                }
                Dft<16>::run_windowed(v, win);
            }
            passes_after_first(v, std::true_type{});
            cf zm[F::NPAIR], zmid;
            {
                const cf* const pb = lds + lds_pad(NC - t);
#pragma unroll
                for (int p = 0; p < F::NPAIR; ++p) {
                    const cf z = pb[-lds_pad_c(p * LPF)];
                    zm[p] = (p == 0 && t == 0) ? v[F::reg_of_spectrum(0)] : z;
                }
                zmid = lds[lds_pad(NC / 2)];
            }
            if constexpr (FUSE) {
#pragma unroll
                for (int q = 0; q < MELQ; ++q)
                    if (t + LPF * q < fz.n_mels) grow[t + LPF * q] = gq[q];
            }
            wave_lds_fence();                                   // partners in registers: the exchange area is free again

            // ---- gradient spectrum, pair by pair -> operands of the inverse transform
            cf ptw[F::NPAIR];
            {
                const f4* pl = reinterpret_cast<const f4*>(ptwl) + t;
#pragma unroll
                for (int u = 0; u < F::NPAIR / 2; ++u) {
                    const f4 x = pl[u * LPF];
                    ptw[2 * u] = mkc(x.x, x.y);
                    ptw[2 * u + 1] = mkc(x.z, x.w);
                }
            }
            auto bin_grad = [&](int k) {                        // (grad_mel . fb^T)[k]
                const AdjEntry e = adj_lds[k];
                return __builtin_fmaf(e.w0, grow[e.b0], e.w1 * grow[e.b1]);
            };
            cf u[F::NPAIR];
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int k = t + p * LPF;
                cf xk, xm;                                      // X[k], X[NC - k] (scale folded into the window)
                F::r2c_split_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], xk, xm);
                float gkp, gmp;
                if constexpr (FUSE) {
                    gkp = bin_grad(k);
                    gmp = bin_grad(NC - k);
                } else {
                    gkp = gk[p];
                    gmp = gm[p];
                }
                cf hk = norm_pow_grad<POW2>(xk, gkp, power);
                cf hm = norm_pow_grad<POW2>(xm, gmp, power);
                if (p == 0) {                                   // DC and Nyquist: H = 2 Re G
                    const bool dc = (t == 0);
                    hk = mkc(dc ? 2.0f * hk.x : hk.x, dc ? 0.0f : hk.y);
                    hm = mkc(dc ? 2.0f * hm.x : hm.x, dc ? 0.0f : hm.y);
                }
                u[p] = c2r_operand(hk, hm, ptw[p]);                                      // operand k: this lane, register p
                lds[lds_pad(NC - k)] = c2r_operand(hm, hk, mkc(-ptw[p].x, ptw[p].y));    // operand NC - k: lane LPF - t, register 15 - p
                if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (t == 0) {                                       // k = NC / 2 pairs with itself
                cf xk, xm;
                const cf wq = mkc(0.0f, -1.0f);
                F::r2c_split_x2(zmid, zmid, wq, xk, xm);
                float gmid;
                if constexpr (FUSE) gmid = bin_grad(NC / 2);
                else gmid = gmid_reg;
                lds[lds_pad(NC / 2)] = c2r_operand(norm_pow_grad<POW2>(xk, gmid, power), norm_pow_grad<POW2>(xm, gmid, power), wq);
            }
            wave_lds_fence();
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) v[p] = u[p];
            {
                const cf* const src = lds + lds_pad(t);
#pragma unroll
                for (int q = F::NPAIR; q < E; ++q) v[q] = src[lds_pad_c(q * LPF)];
            }
            // ---- inverse transform: R[] in natural order at lds[lds_pad(i)]
            F::template pass_butterflies<0>(v);
            passes_after_first(v, std::false_type{});

            // ---- the group's next frames go out now (v is dead), they land during the epilogue
            if (i + 1 < len) {
                const int fn = (i + 1 < len_lane) ? f0 + i + 1 : f0;
                __builtin_amdgcn_sched_barrier(0);
                request(row, fn);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- windowed frame gradient onto the ring; complete positions out
            bool direct = false;
            if (plan.direct && (sidx == 0 || (f - f0) * hop >= N - hop)) {
                const int jlo = f * hop - g.center_pad, jhi = jlo + hop - 1, L = (int)g.length;
                direct = (g.center_pad == 0 || g.pad_mode == PAD_CONSTANT) ? (jlo >= 0 && jhi < L)
                                                                          : (jlo > g.center_pad && jhi < L - 1 - g.center_pad);
            }
            float* const prow = gpad + (long long)row * plan.pad_len + (long long)f * hop;           // position f·hop
            float* const drow = direct ? plan.gwave + (long long)row * plan.gstride + ((long long)f * hop - g.center_pad) : prow;
            float* const tail = row_end ? prow : tail0;
            {
                const f4* wp = reinterpret_cast<const f4*>(wlds + t * WROW);
                const cf* const src = lds + lds_pad(t);
                cf acc[E];
#pragma unroll
                for (int uu = 0; uu < E / 2; ++uu) {
                    const f4 x = wp[uu];
                    const cf r0 = src[lds_pad_c((2 * uu) * LPF)], r1 = src[lds_pad_c((2 * uu + 1) * LPF)];
                    acc[2 * uu] = cmul_elem(r0, mkc(x.x, -x.y));                 // (Re, -Im) R[m] · window / 2
                    acc[2 * uu + 1] = cmul_elem(r1, mkc(x.z, -x.w));
                }
#pragma unroll
                for (int j = 0; j < R; ++j) acc[j] = cadd(acc[j], ring[j]);
                if (live) {
#pragma unroll
                    for (int j = 0; j < H; ++j) {                                 // complete
                        __builtin_nontemporal_store(acc[j], reinterpret_cast<cf*>(drow + 2 * (t + j * LPF)));
                    }
                    if (last) {                                                   // the segment's open positions
#pragma unroll
                        for (int j = H; j < E; ++j) *reinterpret_cast<cf*>(tail + 2 * (t + j * LPF)) = acc[j];
                    }
                }
#pragma unroll
                for (int j = H; j < E; ++j) ring[j - H] = live ? acc[j] : ring[j - H];
            }
            wave_lds_fence();
        }
    }
}

}  // namespace tac
