// stft_smooth.hip — fft_lengths that are not powers of two (round 5; reference functional.py:99-107 takes any fft_length: 480, 960,
// 1200, 1920 ... are the 10 / 20 / 25 / 40 ms windows of 48 kHz audio, 882 = 20 ms at 44.1 kHz): any EVEN N <= 8192 whose half
// M = N / 2 is 7-smooth, as an M-point complex Stockham transform of radix 4 / 2 / 3 / 5 / 7 passes + the real-input split.
//
// M <= 1024: one frame per WAVE (four per 256-thread workgroup, each wave walking its own frames with wave-level fences between
// the passes — no barrier); larger M: one frame per workgroup.  A frame ping-pongs between two LDS buffers; the N twiddles
// exp(-2 pi i k / N) sit in LDS too (rounded once from double on the host).  Interior
// frames are loaded as 8-byte sample pairs, frames touching the padding sample by sample; rows leave as 8-byte (complex) or
// 4-byte (|X|^p, dB) stores of consecutive bins.  The same plain design as the float64 chain (chain_f64.hip) — these sizes are
// not the measured path; before round 5 they ran as a windowed-DFT matrix product on the fp32 matrix cores (O(N^2): 2 - 16 x
// slower than torch.stft on the same GPU, tools/r05/time_nonpow2.py), which remains the route of odd and non-smooth lengths.
// fft_length 400 keeps its own kernel (stft_n400.hip).
#include "host_common.hpp"

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace tac {
namespace {

constexpr int SM_THREADS = 256;

// radices of the M-point transform: 4s, then a 2, then 3s, 5s, 7s — so the counter of a radix-4 / radix-2 pass is a power of two
struct SmoothPlan {
    int n;
    int r[16];
};

// (powers of two have their own forward kernels; the adjoint of fft_length 8192 is the one place they pass through here)
bool smooth_plan(int n_fft, SmoothPlan* plan, bool allow_pow2 = false) {
    plan->n = 0;
    if (n_fft < 8 || (n_fft & 1) || n_fft > 8192 || (is_pow2(n_fft) && !allow_pow2)) return false;
    int m = n_fft / 2;
    const int radices[5] = {4, 2, 3, 5, 7};
    for (int r : radices)
        while (m % r == 0 && plan->n < 16) {
            plan->r[plan->n++] = r;
            m /= r;
        }
    return m == 1 && plan->n > 0;
}

int smooth_twiddles(int n_fft, const cf** out) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, const cf*> cache;
    int dev = 0;
    TAC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(n_fft, dev);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = it->second;
        return TAC_OK;
    }
    std::vector<cf> host((size_t)n_fft);                                     // all N roots: a lookup is one LDS read, no wrap
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < n_fft; ++k) {
        const double a = -two_pi * (double)k / (double)n_fft;
        float c = (float)std::cos(a), s = (float)std::sin(a);
        if ((4LL * k) % n_fft == 0) {                                        // exact quarter turns stay exact
            const int quad = (int)((4LL * k) / n_fft) & 3;
            c = quad == 0 ? 1.0f : (quad == 2 ? -1.0f : 0.0f);
            s = quad == 1 ? -1.0f : (quad == 3 ? 1.0f : 0.0f);
        }
        host[k] = mkc(c, s);
    }
    cf* dptr = nullptr;
    TAC_HIP(hipMalloc((void**)&dptr, host.size() * sizeof(cf)));
    TAC_HIP(hipMemcpy(dptr, host.data(), host.size() * sizeof(cf), hipMemcpyHostToDevice));
    cache[key] = dptr;
    *out = dptr;
    return TAC_OK;
}

// j mod n for 0 <= j < 2^22 and a pass-constant n: one multiply by the reciprocal and a fix-up instead of an integer division
__device__ __forceinline__ int fast_mod(int j, int n, float inv_n) {
    int q = (int)((float)j * inv_n);
    int r = j - q * n;
    r = r < 0 ? r + n : r;
    return r >= n ? r - n : r;
}

// one radix-3 / 5 / 7 Stockham pass over a frame: the small transform's roots W_R^(t u) = W_N^((N / R)(t u mod R)) come from the
// table
template <int R, int TPF, class Twiddle>
__device__ __forceinline__ void smooth_pass_odd(const cf* __restrict__ src, cf* __restrict__ dst, int M, int N, int ns, int lt,
                                                Twiddle wn) {
    const int cnt = M / R, step = cnt / ns, unit_r = N / R;
    const float inv_ns = 1.0f / (float)ns;
    cf root[R];                                                              // W_R^t, t < R (root[0] unused)
#pragma unroll
    for (int t = 1; t < R; ++t) root[t] = wn(unit_r * t);
    for (int j = lt; j < cnt; j += TPF) {
        const int k = fast_mod(j, ns, inv_ns), q = k * step;
        cf v[R];
        v[0] = src[j];
#pragma unroll
        for (int t = 1; t < R; ++t) v[t] = cmul(src[j + t * cnt], wn(2 * t * q));
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            cf acc = v[0];
#pragma unroll
            for (int t = 1; t < R; ++t) acc = (t * u) % R == 0 ? cadd(acc, v[t]) : cadd(acc, cmul(v[t], root[(t * u) % R]));
            dst[j0 + u * ns] = acc;
        }
    }
}

// the M-point forward transform of one frame: Stockham passes from `bufa`, ping-ponging with `bufb`; returns the buffer that holds
// the spectrum in natural order.  `sync` separates the passes (a wave-level fence at one frame per wave, else a barrier).
template <int TPF, class Twiddle, class Sync>
__device__ __forceinline__ cf* smooth_transform(cf* bufa, cf* bufb, const SmoothPlan& plan, int M, int N, int lt, Twiddle wn,
                                                Sync sync) {
    cf* src = bufa;
    cf* dst = bufb;
    int ns = 1;
    for (int p = 0; p < plan.n; ++p) {                                    // Stockham passes, ns = product of the radices so far
        const int r = plan.r[p], cnt = M / r, step = cnt / ns;            // pass twiddle W_M^(t k step) = W_N^(2 t k step)
        if (r == 4 && ns == 1) {                                          // first pass: every twiddle is 1
#pragma unroll 2
            for (int j = lt; j < cnt; j += TPF) {
                const cf v0 = src[j], v1 = src[j + cnt], v2 = src[j + 2 * cnt], v3 = src[j + 3 * cnt];
                const cf s0c = cadd(v0, v2), s1c = csub(v0, v2), s2c = cadd(v1, v3), s3c = csub(v1, v3);
                cf* const d = dst + 4 * j;
                d[0] = cadd(s0c, s2c);
                d[1] = mkc(s1c.x + s3c.y, s1c.y - s3c.x);
                d[2] = csub(s0c, s2c);
                d[3] = mkc(s1c.x - s3c.y, s1c.y + s3c.x);
            }
        } else if (r == 4) {
#pragma unroll 2
            for (int j = lt; j < cnt; j += TPF) {
                const int k = j & (ns - 1), q = k * step;
                const cf v0 = src[j], v1 = cmul(src[j + cnt], wn(2 * q)), v2 = cmul(src[j + 2 * cnt], wn(4 * q)),
                         v3 = cmul(src[j + 3 * cnt], wn(6 * q));
                const cf s0c = cadd(v0, v2), s1c = csub(v0, v2), s2c = cadd(v1, v3), s3c = csub(v1, v3);
                const int j0 = ((j - k) << 2) + k;
                dst[j0] = cadd(s0c, s2c);
                dst[j0 + ns] = mkc(s1c.x + s3c.y, s1c.y - s3c.x);         // s1 - i s3
                dst[j0 + 2 * ns] = csub(s0c, s2c);
                dst[j0 + 3 * ns] = mkc(s1c.x - s3c.y, s1c.y + s3c.x);
            }
        } else if (r == 2) {
#pragma unroll 2
            for (int j = lt; j < cnt; j += TPF) {
                const int k = j & (ns - 1);
                const cf v0 = src[j], v1 = cmul(src[j + cnt], wn(2 * k * step));
                const int j0 = ((j - k) << 1) + k;
                dst[j0] = cadd(v0, v1);
                dst[j0 + ns] = csub(v0, v1);
            }
        } else if (r == 3) {
            smooth_pass_odd<3, TPF>(src, dst, M, N, ns, lt, wn);
        } else if (r == 5) {
            smooth_pass_odd<5, TPF>(src, dst, M, N, ns, lt, wn);
        } else {
            smooth_pass_odd<7, TPF>(src, dst, M, N, ns, lt, wn);
        }
        sync();
        cf* t = src; src = dst; dst = t;
        ns *= r;
    }
    return src;
}

// MODE 0: complex rows [F][2]; 1: |X|^power rows [F] (+ dB).  TPF: threads per frame — 64 (M <= 1024: one frame per wave, the four
// waves of a workgroup walk their own frames and only wave-level fences separate the passes) or 256 (one frame per workgroup,
// barriers).  LDS: [N twiddles][256 / TPF frames][2][M].
template <int MODE, int TPF, bool FULL = true>
__global__ void __launch_bounds__(SM_THREADS)
stft_smooth_kernel(FrameGeom g, const cf* __restrict__ tw, StftEpilogue ep, SmoothPlan plan, int N, int win_vec2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int SLOTS = SM_THREADS / TPF;
    cf* const wl = reinterpret_cast<cf*>(smem_raw);
    const int M = N >> 1, tid = threadIdx.x, lt = tid % TPF;
    const int slot = __builtin_amdgcn_readfirstlane(tid / TPF);
    // FULL: all N roots in LDS (a lookup is one read); else the first M + 1 and W^(j) = -W^(j - M) — the largest sizes, where the
    // full table would leave room for one workgroup per CU only
    cf* const bufa = wl + (FULL ? N : M + 1) + slot * 2 * M;
    cf* const bufb = bufa + M;
    for (int i = tid; i < (FULL ? N : M + 1); i += SM_THREADS) wl[i] = tw[i];
    __syncthreads();
    auto wn = [&](int j) -> cf {                                             // exp(-2 pi i j / N), 0 <= j < N
        if constexpr (FULL) return wl[j];
        const int r = j >= M ? j - M : j;
        const cf v = wl[r];
        return j >= M ? mkc(-v.x, -v.y) : v;
    };
    auto sync = [&]() {
        if constexpr (TPF == 64) wave_lds_fence();
        else __syncthreads();
    };
    const int F = ep.onesided ? M + 1 : N;
    const long long T = g.n_frames, units = g.rows * T;
    const int L = (int)g.length;
    for (long long unit = (long long)blockIdx.x * SLOTS + slot; unit < units; unit += (long long)gridDim.x * SLOTS) {
        const long long row = unit / T;
        const float* __restrict__ rp = g.wave + row * g.row_stride;
        const long long s0 = (unit - row * T) * g.hop - g.center_pad;
        if (g.vec2_ok && s0 >= 0 && s0 + N <= g.length) {                     // (uniform per frame)
            const cf* __restrict__ xp = reinterpret_cast<const cf*>(rp + s0);
            if (win_vec2) {
                const cf* __restrict__ wp = reinterpret_cast<const cf*>(g.window);
#pragma unroll 4
                for (int i = lt; i < M; i += TPF) bufa[i] = cmul_elem(xp[i], wp[i]);
            } else {
#pragma unroll 2
                for (int i = lt; i < M; i += TPF) bufa[i] = cmul_elem(xp[i], window_pair(g, i));
            }
        } else {
            for (int i = lt; i < M; i += TPF) {
                bool z0, z1;
                const int j0 = padded_index((int)s0 + 2 * i, L, g.pad_mode, &z0);
                const int j1 = padded_index((int)s0 + 2 * i + 1, L, g.pad_mode, &z1);
                const float a = rp[j0], b = rp[j1];
                bufa[i] = cmul_elem(mkc(z0 ? 0.0f : a, z1 ? 0.0f : b), window_pair(g, i));
            }
        }
        sync();
        cf* const src = smooth_transform<TPF>(bufa, bufb, plan, M, N, lt, wn, sync);
        // real-input split: X[k] = (Z[k] + conj Z[M-k]) / 2 - i W_N^k (Z[k] - conj Z[M-k]) / 2, k = 0..M
#pragma unroll 2
        for (int k = lt; k <= M; k += TPF) {
            const cf a = src[k == M ? 0 : k], braw = src[k == 0 ? 0 : M - k];
            const cf b = mkc(braw.x, -braw.y);
            const cf e = cscale(cadd(a, b), 0.5f), o = cscale(csub(a, b), 0.5f);
            const cf wo = cmul(wn(k), o);                                    // (W_N^M = -1 is in the table)
            const cf X = cscale(mkc(e.x + wo.y, e.y - wo.x), g.scale);       // e - i (w o)
            if constexpr (MODE == 0) {
                cf* const o2 = reinterpret_cast<cf*>(ep.out) + unit * F;
                o2[k] = X;
                if (!ep.onesided && k > 0 && k < M) o2[N - k] = mkc(X.x, -X.y);
            } else {
                const float s = cnorm2(X);
                float val = (ep.power == 2.0f) ? s : ((ep.power == 1.0f) ? sqrtf(s) : powf(sqrtf(s), ep.power));
                if (ep.db) val = amp_to_db(val, ep.amin, ep.log10_ref);
                float* const o1 = ep.out + unit * F;
                o1[k] = val;
                if (!ep.onesided && k > 0 && k < M) o1[N - k] = val;
            }
        }
        sync();                                                               // the next frame's deposit follows these reads
    }
}

// Adjoint of the forward kernel up to the overlap-add (tac_stft_backward_f32): grad_frames[unit][n] = window[n] * scale *
// Re sum_{k <= M} G[k] e^{+2 pi i k n / N}.  With n = 2 m + b the two real M-point inverse transforms pack into one complex one,
//     Z[k] = (G[k] + conj G[M-k]) / 2 + (i / 2) conj(W_N^k) (G[k] - conj G[M-k])   (0 < k < M),   Z[0] = Re(G[0] + G[M]) + i Re(G[0] - G[M]),
//     z = IDFT_M(Z) = conj(DFT_M(conj Z)),  y[2m] = Re z[m],  y[2m+1] = Im z[m]
// — the forward passes run on conj Z and the result is conjugated on the way out.
template <int TPF, bool FULL = true>
__global__ void __launch_bounds__(SM_THREADS)
stft_smooth_backward_kernel(FrameGeom g, const cf* __restrict__ tw, const cf* __restrict__ grad_spec, float* __restrict__ grad_frames,
                            SmoothPlan plan, int N) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int SLOTS = SM_THREADS / TPF;
    cf* const wl = reinterpret_cast<cf*>(smem_raw);
    const int M = N >> 1, tid = threadIdx.x, lt = tid % TPF;
    const int slot = __builtin_amdgcn_readfirstlane(tid / TPF);
    cf* const bufa = wl + (FULL ? N : M + 1) + slot * 2 * M;
    cf* const bufb = bufa + M;
    for (int i = tid; i < (FULL ? N : M + 1); i += SM_THREADS) wl[i] = tw[i];
    __syncthreads();
    auto wn = [&](int j) -> cf {
        if constexpr (FULL) return wl[j];
        const int r = j >= M ? j - M : j;
        const cf v = wl[r];
        return j >= M ? mkc(-v.x, -v.y) : v;
    };
    auto sync = [&]() {
        if constexpr (TPF == 64) wave_lds_fence();
        else __syncthreads();
    };
    const long long units = g.rows * g.n_frames;
    for (long long unit = (long long)blockIdx.x * SLOTS + slot; unit < units; unit += (long long)gridDim.x * SLOTS) {
        const cf* __restrict__ G = grad_spec + unit * (M + 1);
#pragma unroll 2
        for (int k = lt; k < M; k += TPF) {
            cf zc;                                                            // conj Z[k]
            if (k == 0) {
                const cf g0 = G[0], gm = G[M];
                zc = mkc(g0.x + gm.x, -(g0.x - gm.x));
            } else {
                const cf a = G[k], braw = G[M - k];
                const cf b = mkc(braw.x, -braw.y);
                const cf e = cscale(cadd(a, b), 0.5f), o = cscale(csub(a, b), 0.5f);
                const cf w = wn(k);
                const cf wo = cmul(mkc(w.x, -w.y), o);                        // conj(W_N^k) o
                zc = mkc(e.x - wo.y, -(e.y + wo.x));                          // conj(e + i wo)
            }
            bufa[k] = zc;
        }
        sync();
        const cf* const res = smooth_transform<TPF>(bufa, bufb, plan, M, N, lt, wn, sync);
        float* const out = grad_frames + unit * N;
#pragma unroll 2
        for (int m = lt; m < M; m += TPF) {
            const cf v = res[m], w = window_pair(g, m);
            *reinterpret_cast<cf*>(out + 2 * m) = mkc(v.x * w.x * g.scale, -v.y * w.y * g.scale);
        }
        sync();                                                               // the next frame's deposit follows these reads
    }
}

}  // namespace

bool stft_smooth_covers(int n_fft) {
    SmoothPlan plan;
    return n_fft != 400 && smooth_plan(n_fft, &plan);
}

// Entry used by stft_kernels.hip's dispatcher (mode 0: complex rows, 1: |X|^power rows with the optional dB epilogue).
int launch_stft_smooth(int n_fft, const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream) {
    SmoothPlan plan;
    if (!smooth_plan(n_fft, &plan)) return TAC_E_UNSUPPORTED;
    const cf* tw = nullptr;
    const int rc = smooth_twiddles(n_fft, &tw);
    if (rc != TAC_OK) return rc;
    const int M = n_fft / 2;
    const long long units = g.rows * g.n_frames;
    const int tpf = M <= 1024 ? 64 : 256, slots = SM_THREADS / tpf;
    const size_t lds_full = ((size_t)n_fft + 2 * (size_t)M * slots) * sizeof(cf), lds_half = lds_full - (size_t)(M - 1) * sizeof(cf);
    const bool full = !((160 * 1024) / lds_full < 2 && (160 * 1024) / lds_half >= 2);       // (6000: 96 KB against 72 KB)
    const size_t lds = full ? lds_full : lds_half;
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    long long blocks = (units + slots - 1) / slots;
    const long long cap = (long long)device_cu_count() * per_cu;
    if (blocks > cap) blocks = cap;
    const int win_vec2 = g.win_length == n_fft && (reinterpret_cast<uintptr_t>(g.window) & 7u) == 0;
    auto go = [&](auto kern) -> int {
        TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SM_THREADS), lds, stream, g, tw, ep, plan, n_fft, win_vec2);
        return TAC_OK;
    };
    const int rl = mode == 0 ? (tpf == 64 ? go(stft_smooth_kernel<0, 64>) : (full ? go(stft_smooth_kernel<0, 256>) : go(stft_smooth_kernel<0, 256, false>)))
                             : (tpf == 64 ? go(stft_smooth_kernel<1, 64>) : (full ? go(stft_smooth_kernel<1, 256>) : go(stft_smooth_kernel<1, 256, false>)));
    if (rl != TAC_OK) return rl;
    TAC_HIP(hipGetLastError());
    set_last_route("stft_smooth_kernel<%d, %d>", mode == 0 ? 0 : 1, tpf);
    return TAC_OK;
}

// Frame gradients from a one-sided gradient spectrum (stft_backward_entry of backward.hip): grad_spec[rows][T][M + 1][2] ->
// grad_frames[rows][T][N].
int launch_stft_smooth_backward(int n_fft, const FrameGeom& g, const float* grad_spec, float* grad_frames, hipStream_t stream) {
    SmoothPlan plan;
    if (!smooth_plan(n_fft, &plan, n_fft == 8192)) return TAC_E_UNSUPPORTED;
    const cf* tw = nullptr;
    const int rc = smooth_twiddles(n_fft, &tw);
    if (rc != TAC_OK) return rc;
    const int M = n_fft / 2;
    const long long units = g.rows * g.n_frames;
    const int tpf = M <= 1024 ? 64 : 256, slots = SM_THREADS / tpf;
    const size_t lds_full = ((size_t)n_fft + 2 * (size_t)M * slots) * sizeof(cf), lds_half = lds_full - (size_t)(M - 1) * sizeof(cf);
    const bool full = !((160 * 1024) / lds_full < 2 && (160 * 1024) / lds_half >= 2);
    const size_t lds = full ? lds_full : lds_half;
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    long long blocks = (units + slots - 1) / slots;
    const long long cap = (long long)device_cu_count() * per_cu;
    if (blocks > cap) blocks = cap;
    auto go = [&](auto kern) -> int {
        TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SM_THREADS), lds, stream, g, tw, reinterpret_cast<const cf*>(grad_spec),
                           grad_frames, plan, n_fft);
        return TAC_OK;
    };
    const int rl = tpf == 64 ? go(stft_smooth_backward_kernel<64>)
                             : (full ? go(stft_smooth_backward_kernel<256>) : go(stft_smooth_backward_kernel<256, false>));
    if (rl != TAC_OK) return rl;
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // namespace tac
