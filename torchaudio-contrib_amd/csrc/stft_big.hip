// stft_big.hip — fft_length 8192 / 16384 / 32768 (round 5; reference functional.py:99-107 takes any fft_length): one frame per
// WORKGROUP as a four-step transform over the wave-level 1024-point FFT of fft_core.hpp.
//
// A frame of N = 2048 S real samples (S = 4, 8, 16) is M = 1024 S complex points z[n] = x[2n] + i x[2n+1]; decimated S ways,
//     A_j = FFT_1024(z[S m + j])                         (wave j of the S-wave workgroup, WaveFft<1024, 16> as everywhere else)
//     Z[k + 1024 q] = sum_j W_S^{jq} (W_M^{jk} A_j[k])    (one S-point in-register DFT per column k, Dft<S>)
//     X[k] = (ev + W_N^k (-i d)) / 2,  X[M - k] = conj(ev - W_N^k (-i d)) / 2,  ev / d = Z[k] +- conj(Z[M - k])       (R2C split)
// The frame lives in LDS once: S regions of 1024 (padded: 1090) complex slots — region j first receives z[S m + j] (windowed at
// the coalesced 16-byte load), is wave j's exchange area during its transform, then holds A_j, then Z[1024 j ..] (the column DFTs
// are in place: a thread reads and writes the same S slots).  8.7 KB per wave: 35 / 70 / 140 KB per workgroup.  The twiddles of
// a thread's columns (W_M^{jk}) and pairs (W_N^k) do not depend on the frame: loaded once into registers, the workgroups are
// persistent.  Two-sided output mirrors the conjugates; real rows take any power (and the dB epilogue) per element.
// Before round 5 these lengths went through the windowed-DFT matrix product (N <= 8192, O(N^2)) or torch's operators (above).
#include "host_common.hpp"

namespace tac {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int BIG_WS = 1090;          // complex slots per region: WaveFft<1024, 16>::PADDED (1089) rounded up to 16 bytes

template <int S>
__device__ __forceinline__ cf big_z(const cf* smem, int i) { return smem[(i >> 10) * BIG_WS + lds_pad(i & 1023)]; }

// WAVES: waves of the workgroup (S, or S / 2 with two transforms per wave: 16 waves of 128 registers cannot hold the twiddles)
template <int S, int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 8 / WAVES)
stft_big_kernel(FrameGeom g, Tables tb1k, Tables tbn, StftEpilogue ep, int win_vec4) {
    using F = WaveFft<1024, 16>;
    static_assert(F::PADDED <= BIG_WS, "region smaller than the exchange area");
    constexpr int M = 1024 * S, N = 2 * M, NT = 64 * WAVES, LS = S == 4 ? 2 : (S == 8 ? 3 : 4), COLS = 1024 / NT, SUBS = S / WAVES;
    constexpr int CHUNKS = N / 4 / NT, PAIRS = M / 2 / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* const smem = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, t = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* const mine = smem + w * SUBS * BIG_WS;

    cf tw[F::NTW];
    F::load_twiddles(tw, tb1k.w_nc, t);
    // W_M^{jk} of this thread's columns k = tid + NT i and W_N^k of its pairs k = tid + NT i (k < M / 2) do not depend on the
    // frame: kept in registers where they fit (S = 16: 30 + 16 complex values — re-read per frame from the tables, L2 hits)
    constexpr bool HOIST = COLS * (S - 1) + PAIRS <= 24;
    constexpr int UC = HOIST ? COLS : 1, UP = HOIST ? PAIRS : 4;          // (unroll factors of the column / pair loops)
    cf cw[HOIST ? COLS : 1][S - 1];
    cf pw[HOIST ? PAIRS : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int i = 0; i < COLS; ++i)
#pragma unroll
            for (int j = 1; j < S; ++j) cw[i][j - 1] = tbn.w_nc[j * (tid + NT * i)];
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) pw[i] = tbn.w_n[tid + NT * i];
    }

    const long long T = g.n_frames, total = g.rows * T;
    const int nbins = ep.onesided ? M + 1 : N;
    const int L = (int)g.length;
    const float hs = 0.5f * g.scale;
    // Interior frames are requested one frame ahead (CHUNKS 16-byte loads per thread, issued before the previous frame's row
    // stores) and deposited after them: they land while that frame is split and stored.  The loop body is (B)(C) of the
    // deposited frame, the request of the next, (D), the deposit (A) of the next — so that the request's registers live inside
    // one iteration.
    f4 raw[CHUNKS];
    auto request = [&](long long u) -> bool {
        const long long urow = u / T;
        const long long us0 = (u - urow * T) * g.hop - g.center_pad;
        const bool ok = g.vec4_ok && win_vec4 && us0 >= 0 && us0 + N <= g.length;  // (workgroup-uniform)
        if (ok) {
            const f4* const src = reinterpret_cast<const f4*>(g.wave + urow * g.row_stride + us0);
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) raw[i] = src[tid + NT * i];
        }
        return ok;
    };
    // (A) a frame, windowed, into its decimated places: chunk c = samples 4c .. 4c+3 = z[2c], z[2c+1]
    auto deposit = [&](int c, f4 x, f4 wv) {
        const int region = (2 * c) & (S - 1), m = (2 * c) >> LS;
        cf* const dst = smem + region * BIG_WS + lds_pad(m);
        dst[0] = mkc(x.x * wv.x, x.y * wv.y);
        dst[BIG_WS] = mkc(x.z * wv.z, x.w * wv.w);
    };
    auto fill = [&](long long u, bool requested) {
        if (requested) {
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) {
                deposit(tid + NT * i, raw[i], reinterpret_cast<const f4*>(g.window)[tid + NT * i]);     // (full-length, aligned window)
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // (four window requests in flight, not CHUNKS)
            }
        } else {
            // frames touching the padding, unaligned hops, short or unaligned windows: gathered sample by sample
            const long long urow = u / T;
            const int s0 = (int)((u - urow * T) * g.hop - g.center_pad);
            const float* const rp = g.wave + urow * g.row_stride;
#pragma unroll 2
            for (int i = 0; i < CHUNKS; ++i) {
                const int c = tid + NT * i;
                float smp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bool zero;
                    const int j = padded_index(s0 + 4 * c + e, L, g.pad_mode, &zero);
                    const float vs = rp[j];
                    smp[e] = zero ? 0.0f : vs;
                }
                const cf wa = window_pair(g, 2 * c), wb = window_pair(g, 2 * c + 1);
                deposit(c, f4{smp[0], smp[1], smp[2], smp[3]}, f4{wa.x, wa.y, wb.x, wb.y});
            }
        }
    };
    if ((long long)blockIdx.x < total) fill(blockIdx.x, request(blockIdx.x));
    __syncthreads();
    for (long long unit = blockIdx.x; unit < total; unit += gridDim.x) {
        const long long nxt = unit + gridDim.x;
        // (B) wave w: A_j = FFT_1024 of region j = SUBS w + u, in place (natural order at lds_pad(k))
#pragma unroll
        for (int u = 0; u < SUBS; ++u) {
            cf v[1][16];
            cf* const reg = mine + u * BIG_WS;
#pragma unroll
            for (int q = 0; q < 16; ++q) v[0][q] = reg[lds_pad(t + 64 * q)];
            wave_lds_fence();
            cf* const la[1] = {reg};
            F::template run<1>(v, la, tw, t);
        }
        __syncthreads();
        // (C) columns: Z[k + 1024 q] = DFT_S over j of W_M^{jk} A_j[k], in place
#pragma unroll UC
        for (int i = 0; i < COLS; ++i) {
            cf* const col = smem + lds_pad(tid + NT * i);
            cf a[S];
#pragma unroll
            for (int j = 0; j < S; ++j) a[j] = col[j * BIG_WS];
#pragma unroll
            for (int j = 1; j < S; ++j) a[j] = cmul(a[j], HOIST ? cw[i][j - 1] : tbn.w_nc[j * (tid + NT * i)]);
            Dft<S>::run(a);
#pragma unroll
            for (int q = 0; q < S; ++q) col[q * BIG_WS] = a[q];
        }
        __syncthreads();
        // the next frame's samples, ahead of this frame's stores
        __builtin_amdgcn_sched_barrier(0);
        const bool requested = nxt < total ? request(nxt) : false;
        __builtin_amdgcn_sched_barrier(0);
        // (D) R2C split of the pairs (k, M - k) and the row
        float* const obase = ep.out + unit * (long long)nbins * (MODE == 0 ? 2 : 1);
        auto emit = [&](int bin, cf xv) {
            if constexpr (MODE == 0) {
                cf* const o2 = reinterpret_cast<cf*>(obase);
                o2[bin] = xv;
                if (!ep.onesided && bin > 0 && bin < M) o2[N - bin] = mkc(xv.x, -xv.y);
            } else {
                const float s = cnorm2(xv);
                float val = (ep.power == 2.0f) ? s : ((ep.power == 1.0f) ? sqrtf(s) : powf(sqrtf(s), ep.power));
                if (ep.db) val = amp_to_db(val, ep.amin, ep.log10_ref);
                obase[bin] = val;
                if (!ep.onesided && bin > 0 && bin < M) obase[N - bin] = val;
            }
        };
        auto split = [&](cf zk, cf zm, cf wk, cf& xk, cf& xm) {
            const cf zc = mkc(zm.x, -zm.y);
            const cf ev = cadd(zk, zc), d = csub(zk, zc);
            const cf tt = cmul(mul_neg_i(d), wk);
            xk = cscale(cadd(ev, tt), hs);
            const cf u = csub(ev, tt);
            xm = cscale(mkc(u.x, -u.y), hs);
        };
#pragma unroll UP
        for (int i = 0; i < PAIRS; ++i) {
            const int k = tid + NT * i;
            const cf zk = big_z<S>(smem, k), zm = big_z<S>(smem, (M - k) & (M - 1));
            cf xk, xm;
            split(zk, zm, HOIST ? pw[i] : tbn.w_n[k], xk, xm);
            emit(k, xk);
            emit(M - k, xm);
        }
        if (tid == 0) {
            const cf zh = big_z<S>(smem, M / 2);
            cf xk, xm;
            split(zh, zh, mkc(0.0f, -1.0f), xk, xm);                 // W_N^(M/2) = -i: X[M/2] = conj(Z[M/2])
            emit(M / 2, xk);
        }
        __syncthreads();                                   // the next frame's deposits follow these reads
        if (nxt < total) fill(nxt, requested);
        __syncthreads();
    }
}

template <int S, int MODE>
static int launch_big(const FrameGeom& g, const StftEpilogue& ep, hipStream_t stream) {
    Tables tb1k, tbn;
    int rc = get_tables(2048, &tb1k);
    if (rc != TAC_OK) return rc;
    rc = get_tables(2048 * S, &tbn);
    if (rc != TAC_OK) return rc;
    const long long units = g.rows * g.n_frames;
    const int bytes = S * BIG_WS * (int)sizeof(cf);
    constexpr int WAVES = S == 16 ? 8 : S;      // (S = 8 as two 4-wave workgroups per CU: +28 %, tools/ablation/README.md)
    long long blocks = (long long)device_cu_count() * (8 / WAVES);     // eight waves per CU: the twiddles need 256 registers
    if (blocks > units) blocks = units;
    const int win_vec4 = g.win_length == 2048 * S && (reinterpret_cast<uintptr_t>(g.window) & 15u) == 0;
    auto kern = stft_big_kernel<S, MODE, WAVES>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WAVES), bytes, stream, g, tb1k, tbn, ep, win_vec4);
    TAC_HIP(hipGetLastError());
    set_last_route("stft_big_kernel<%d, %d, %d>", S, MODE, WAVES);
    return TAC_OK;
}

// Entry used by stft_kernels.hip's dispatcher (mode 0: complex rows, 1: |X|^power rows with the optional dB epilogue).
int launch_stft_big(int n_fft, const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream) {
    switch (n_fft) {
        case 8192: return mode == 0 ? launch_big<4, 0>(g, ep, stream) : launch_big<4, 1>(g, ep, stream);
        case 16384: return mode == 0 ? launch_big<8, 0>(g, ep, stream) : launch_big<8, 1>(g, ep, stream);
        case 32768: return mode == 0 ? launch_big<16, 0>(g, ep, stream) : launch_big<16, 1>(g, ep, stream);
        default: return TAC_E_UNSUPPORTED;
    }
}

}  // namespace tac
