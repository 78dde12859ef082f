// backward_ring3.hpp — adjoint of the fused Melspectrogram chain at fft_length 2048 with THREE waves per SIMD.
//
// spectrogram_backward_ola_kernel keeps a segment's running overlap-add in an 8 KB LDS ring per wave, which caps the CU at
// eight 250-register waves.  A frame's 16 chunks of 128 samples are ONE sample pair per lane each, so the ring is 16 - H
// register pairs per lane once its rotation is static — it is when H = hop / 128 is a template parameter: chunk j of the
// current frame always meets ring register j, and the update leaves its sum in register j - H for the next frame.  With
// the ring out of the LDS, twelve waves fit (8.7 KB of exchange area each + 37 KB of shared tables), and the kernel is
// built like melspec_stream3_kernel: one frame per wave, window / pass-1 / R2C twiddles re-read from LDS every frame.
//
// Per frame: forward FFT of the windowed frame (lower half of the spectrum stays in registers) -> per PAIR (k, NC - k):
// R2C split, filterbank adjoint (two multiply-adds per bin through the per-bin table, functional.py:183-184 transposed),
// adjoint of |z|^power (functional.py:126-128), both operands of the inverse transform — the one for NC - k belongs to
// lane 64 - t and crosses through the exchange area — -> inverse FFT (a forward FFT on conjugated data) -> window,
// overlap-add into the register ring, complete positions out.  Replaces autograd through layers.py:333-339.
#pragma once
#include "melspec_stream3.hpp"

namespace tac {

constexpr int BR_WAVES = 12;

                               // kernel at cfg-2 (rocprofv3 FETCH_SIZE + WRITE_SIZE): 631 -> 517 (3) -> 448 MB (7), 369 MB compulsory; time
                               // -0.8 % (3), -1.1 % (7), bit-identical (tools/ablation/README.md)
template <class T>
__device__ __forceinline__ T br3_load_once(const T* p) {
    return __builtin_nontemporal_load(p);
}

// Round 6: the pass-2 twiddles with the last pass's W_16 constants multiplied in — W_NC^((t + 64 b) q), b < 4, q = 1 .. 3, as [u < 6][lane]
// pairs in LDS — instead of three hoisted registers and nine constant multiplies per transform: the kernel runs two transforms per frame
// at the 168-register cap with 24 KB of LDS to spare, and is bound by its instruction stream: 36 fewer vector instructions per frame of
// ~965 and five fewer spilled registers for twelve more 16-byte LDS reads: 0.2585 -> 0.2407 ms (-6.9 %, same process,
// profiles/r06/ab/batch23_backward_pass2_table.txt).
constexpr int BR3_TW2L_BYTES = 64 * 12 * (int)sizeof(cf);

template <int NC, int E>
__host__ __device__ inline size_t ring3_lds_bytes(int mel_stride) {
    using F = WaveFft<NC, E>;
    size_t xa = ((size_t)F::PADDED * sizeof(cf) + 15) & ~(size_t)15;
    return (size_t)BR_WAVES * xa + ST_TW_BYTES + 64 * (F::NPAIR + E) * sizeof(cf) + (size_t)(NC + 1) * sizeof(AdjEntry) +
           (size_t)BR_WAVES * mel_stride * sizeof(float) + BR3_TW2L_BYTES;
}

// FUSE: `gmel` is the gradient of the mel values and the filterbank adjoint happens here (fz); otherwise it is the gradient of
// |z|^power itself, (rows, T, NC + 1) frame-major (tac_spectrogram_backward_ola_f32), fetched per pair at the top of the frame
template <bool POW2, int H, bool FUSE>
__global__ void __launch_bounds__(BR_WAVES * 64, 3)
melspec_backward_ring3_kernel(FrameGeom g, Tables tb, const float* __restrict__ gmel, float power, float* __restrict__ gpad,
                              float* __restrict__ edge, OlaPlan plan, OlaFuse fz) {
    constexpr int NC = OLA_NC, E = OLA_E, N = OLA_N, NBINS = NC + 1, R = 16 - H, WAVES = BR_WAVES;
    using F = WaveFft<NC, E>;
    static_assert(H >= 1 && H < 16, "hop = 128 H < fft_length");
    constexpr int XA_BYTES = (F::PADDED * sizeof(cf) + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* const xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);
    float* const twlds = reinterpret_cast<float*>(smem_raw + (size_t)WAVES * XA_BYTES);
    if (tid < 16 * 16) {
        const int js = tid >> 4, q = tid & 15;
        const cf wv = q ? tb.w_nc[js * q * (NC / 256)] : mkc(1.0f, 0.0f);
        twlds[js * ST_TW_STRIDE + 2 * (q ? q - 1 : 15)] = wv.x;
        twlds[js * ST_TW_STRIDE + 2 * (q ? q - 1 : 15) + 1] = wv.y;
    }
    // R2C / C2R twiddles w_k of a lane's eight pairs and its sixteen window pairs as [read u][lane] 16-byte pairs
    cf* const ptwl = reinterpret_cast<cf*>(twlds + ST_TW_BYTES / 4);
    for (int idx = tid; idx < 64 * F::NPAIR; idx += WAVES * 64) {
        const int tt = idx / F::NPAIR, p = idx - tt * F::NPAIR;
        ptwl[((p >> 1) * 64 + tt) * 2 + (p & 1)] = tb.w_n[tt + p * F::LPF];
    }
    const float half = 0.5f * g.scale;      // forward: the R2C split returns 2X; inverse: the common 1/2 of the C2R operands
    cf* const winl = ptwl + 64 * F::NPAIR;
    for (int idx = tid; idx < 64 * E; idx += WAVES * 64) {
        const int tt = idx / E, q = idx - tt * E;
        winl[((q >> 1) * 64 + tt) * 2 + (q & 1)] = cscale(window_pair(g, tt + q * F::LPF), half);
    }
    cf* const tw2l = winl + 64 * E;
    for (int i = tid; i < 64 * 12; i += WAVES * 64) {
        const int tt = i & 63, e = i >> 6, b = e / 3, q = e % 3 + 1;
        tw2l[((e >> 1) * 64 + tt) * 2 + (e & 1)] = tb.w_nc[((tt + 64 * b) * q) & 1023];
    }
    AdjEntry* const adj_lds = reinterpret_cast<AdjEntry*>(reinterpret_cast<unsigned char*>(tw2l) + BR3_TW2L_BYTES);
    if constexpr (FUSE)
        for (int k = tid; k < NBINS; k += WAVES * 64) adj_lds[k] = fz.adj[k];
    float* const grow = reinterpret_cast<float*>(adj_lds + NBINS) + w * fz.mel_stride;
    __syncthreads();

    const int T = (int)g.n_frames, hop = g.hop, S = plan.seg_frames, spr = plan.segs_per_row;
    const long long nseg_total = g.rows * (long long)spr;
    const long long stride = (long long)gridDim.x * WAVES;
    long long seg = (long long)blockIdx.x * WAVES + w;
    if (seg >= nseg_total) return;
    int row = (int)(seg / spr), sidx = (int)(seg - (long long)row * spr);
    int f0 = sidx * S, f1 = f0 + S < T ? f0 + S : T, f = f0;

    typedef float f4 __attribute__((ext_vector_type(4)));
    cf v[E];
    float gq[4];
    int mode = 0;
    auto request = [&](int r, int fr) {                     // samples + mel-gradient row of (row r, frame fr), unconditionally
        if constexpr (FUSE) {
            const float* gn = gmel + ((long long)r * T + fr) * fz.n_mels;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = t + 64 * i;
                gq[i] = br3_load_once(gn + (b < fz.n_mels ? b : fz.n_mels - 1));
            }
        }
        const long long start = (long long)fr * hop - g.center_pad;
        const bool ok = g.vec2_ok && start >= 0 && start + N <= g.length;
        mode = ok ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + N <= g.length ? cs : g.length - N;
        const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)r * g.row_stride + cs);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            if (q < H) v[q] = __builtin_nontemporal_load(src + t + q * F::LPF);      // the frame's oldest hop: no later frame reads it
            else v[q] = src[t + q * F::LPF];
        }
    };
    auto load_tw1 = [&](cf (&tw1)[16]) {
        const f4* tl = reinterpret_cast<const f4*>(twlds + (t & 15) * ST_TW_STRIDE);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f4 x = tl[u];
            tw1[2 * u] = mkc(x.x, x.y);
            tw1[2 * u + 1] = mkc(x.z, x.w);
        }
    };
    // passes 1 and 2 of the 16.16.4 plan on the outputs of pass 0 (in v); HALF: the lower half of the result stays in registers
    auto passes_after_first = [&](auto half_tag) {
        constexpr bool HALF = decltype(half_tag)::value;
        wave_lds_fence();
        cf tw1[16];
        load_tw1(tw1);
        F::template pass_write<0, true>(v, xa, t, t);
        wave_lds_fence();
        s3_readback_pass1<F>(v, xa, t);                     // (single ds_read_b64s, melspec_stream3.hpp)
        F::template pass_twiddle<1, true>(v, tw1);
        F::template pass_butterflies<1>(v);
        F::exchange_1_2_in_registers(v);
        {
            const f4* tl2 = reinterpret_cast<const f4*>(tw2l) + t;
            cf w2[12];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const f4 x = tl2[u * 64];
                w2[2 * u] = mkc(x.x, x.y);
                w2[2 * u + 1] = mkc(x.z, x.w);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                cmul_x2(v[4 * b + 1], w2[3 * b], v[4 * b + 2], w2[3 * b + 1]);
                v[4 * b + 3] = cmul(v[4 * b + 3], w2[3 * b + 2]);
            }
        }
        F::template pass_butterflies<2>(v);
        wave_lds_fence();
        F::template pass_write<2, HALF>(v, xa, t, t);
        wave_lds_fence();
    };

    cf ring[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ring[r] = mkc(0.0f, 0.0f);
    request(row, f);
    while (true) {
        // the item after this one: next frame of the segment, or the first frame of this wave's next segment
        long long nseg = seg;
        int nrow = row, nsidx = sidx, nf0 = f0, nf1 = f1, nf = f + 1;
        bool more = true;
        if (nf >= f1) {
            nseg = seg + stride;
            if (nseg >= nseg_total) {
                more = false;
                nf = f;
            } else {
                nrow = (int)(nseg / spr);
                nsidx = (int)(nseg - (long long)nrow * spr);
                nf0 = nsidx * S;
                nf1 = nf0 + S < T ? nf0 + S : T;
                nf = nf0;
            }
        }
        const bool last = (f + 1 == f1);

        // (unfused: the gradient values of this lane's eight pairs and of bin NC / 2 travel behind the forward transform)
        float gk[F::NPAIR], gm[F::NPAIR], gmid_reg = 0.0f;
        if constexpr (!FUSE) {
            const float* gn = gmel + ((long long)row * T + f) * NBINS;
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                gk[p] = br3_load_once(gn + t + p * F::LPF);
                gm[p] = br3_load_once(gn + NC - (t + p * F::LPF));
            }
            gmid_reg = br3_load_once(gn + NC / 2);
        }
        // ---- forward transform of the frame
        if (mode != 1) {                                    // frames touching the padding gather their samples first
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v, g, nullptr, xa, row, f, tz, FetchF32{g.wave});
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
        passes_after_first(std::true_type{});
        cf zm[F::NPAIR], zmid;
        s3_read_partners<F>(v, xa, zm, zmid, t);
        // the frame's mel-gradient row, for the per-bin gathers
        if constexpr (FUSE) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (t + 64 * i < fz.n_mels) grow[t + 64 * i] = gq[i];
        }
        wave_lds_fence();                                   // partners in registers: the exchange area is free again

        // ---- gradient spectrum, pair by pair -> operands of the inverse transform
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
        auto bin_grad = [&](int k) {                        // (grad_mel . fb^T)[k]
            const AdjEntry e = adj_lds[k];
            return __builtin_fmaf(e.w0, grow[e.b0], e.w1 * grow[e.b1]);
        };
        cf u[F::NPAIR];
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) {
            const int k = t + p * F::LPF;
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
            u[p] = c2r_operand(hk, hm, ptw[p]);                                     // operand k: this lane, register p
            xa[lds_pad(NC - k)] = c2r_operand(hm, hk, mkc(-ptw[p].x, ptw[p].y));    // operand NC - k: lane 64 - t, register 15 - p
            if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (t == 0) {                                       // k = NC / 2 pairs with itself
            cf xk, xm;
            const cf wq = mkc(0.0f, -1.0f);
            F::r2c_split_x2(zmid, zmid, wq, xk, xm);
            float gmid;
            if constexpr (FUSE) gmid = bin_grad(NC / 2);
            else gmid = gmid_reg;
            xa[lds_pad(NC / 2)] = c2r_operand(norm_pow_grad<POW2>(xk, gmid, power), norm_pow_grad<POW2>(xm, gmid, power), wq);
        }
        wave_lds_fence();
#pragma unroll
        for (int p = 0; p < F::NPAIR; ++p) v[p] = u[p];
        s3_read_strided<F::NPAIR, 8>(&v[F::NPAIR], xa + lds_pad(t));
        // ---- inverse transform: R[] in natural order at xa[lds_pad(i)]
        F::template pass_butterflies<0>(v);
        passes_after_first(std::false_type{});

        // ---- the next frame's samples and gradient row go out now (v is dead), they land during the epilogue
        __builtin_amdgcn_sched_barrier(0);
        request(nrow, nf);
        __builtin_amdgcn_sched_barrier(0);

        // ---- windowed frame gradient onto the ring; complete positions out
        // (wave-uniform; ola_direct() with the segment already known: no division)
        bool direct = false;
        if (plan.direct && (sidx == 0 || (f - f0) * hop >= N - hop)) {
            const int jlo = f * hop - g.center_pad, jhi = jlo + hop - 1, L = (int)g.length;
            direct = (g.center_pad == 0 || g.pad_mode == PAD_CONSTANT) ? (jlo >= 0 && jhi < L)
                                                                      : (jlo > g.center_pad && jhi < L - 1 - g.center_pad);
        }
        float* const prow = gpad + (long long)row * plan.pad_len + (long long)f * hop;           // position f·hop
        float* const drow = direct ? plan.gwave + (long long)row * plan.gstride + ((long long)f * hop - g.center_pad) : prow;
        const bool row_end = (f1 == T);
        float* const tail = row_end ? prow : edge + ((long long)row * (spr - 1) + sidx) * (N - hop) - hop;   // + n
        {
            const f4* wl = reinterpret_cast<const f4*>(winl) + t;
            const cf* const src = xa + lds_pad(t);
            cf acc[E];
            s3_read_strided<0, 8>(&acc[0], src);            // (two batches of eight: the kernel sits at its register cap)
#pragma unroll
            for (int uu = 0; uu < E / 4; ++uu) {
                const f4 x = wl[uu * 64];
                acc[2 * uu] = cmul_elem(acc[2 * uu], mkc(x.x, -x.y));        // (Re, -Im) R[m] · window / 2
                acc[2 * uu + 1] = cmul_elem(acc[2 * uu + 1], mkc(x.z, -x.w));
            }
            s3_read_strided<8, 8>(&acc[8], src);
#pragma unroll
            for (int uu = E / 4; uu < E / 2; ++uu) {
                const f4 x = wl[uu * 64];
                acc[2 * uu] = cmul_elem(acc[2 * uu], mkc(x.x, -x.y));
                acc[2 * uu + 1] = cmul_elem(acc[2 * uu + 1], mkc(x.z, -x.w));
            }
#pragma unroll
            for (int j = 0; j < R; ++j) acc[j] = cadd(acc[j], ring[j]);
#pragma unroll
            for (int j = 0; j < H; ++j) {                                     // complete
                __builtin_nontemporal_store(acc[j], reinterpret_cast<cf*>(drow + 2 * (t + j * 64)));
            }
            if (last) {                                                       // the segment's open positions
#pragma unroll
                for (int j = H; j < E; ++j) *reinterpret_cast<cf*>(tail + 2 * (t + j * 64)) = acc[j];
            }
            const bool fresh = (nf == nf0);                                   // the next item starts a segment: empty ring
#pragma unroll
            for (int j = H; j < E; ++j) ring[j - H] = fresh ? mkc(0.0f, 0.0f) : acc[j];
        }
        wave_lds_fence();
        if (!more) break;
        seg = nseg; row = nrow; sidx = nsidx; f0 = nf0; f1 = nf1; f = nf;
    }
}

}  // namespace tac
