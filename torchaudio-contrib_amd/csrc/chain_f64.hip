// chain_f64.hip — the path in float64 (the reference keeps f64 -> f64: functional.py:48-113 stft, :116-128 complex_norm,
// :172-184 apply_filterbank, :277-296 amplitude_to_db, :299-314 db_to_amplitude, :187-201 angle / magphase).
//
// float64 is not the measured hot path (north_star is float32) — these kernels exist so that a float64 tensor on a gfx950
// device is evaluated by this library like a float32 one, not by vendor FFT / BLAS calls.  Design: one 256-thread
// workgroup per frame, the frame's N/2-point complex Stockham transform (radix 4 / 2 / 3 / 5 passes) ping-ponging
// between two LDS buffers, the N/2 + 1 twiddles exp(-2 pi i k / N) in LDS as well (rounded once from long double on
// the host), real-input split + |X|^p / dB epilogue on the way out; row stores are 16-byte complex pairs.  Odd sizes
// and sizes whose half has a prime factor above 5 take the direct N x F transform from the same tables.  The
// double-precision vector rate of gfx950 is half the float32 one and LDS traffic doubles, so the kernel is LDS- and
// issue-bound by construction; no MFMA (v_mfma_f64 peak equals the vector FMA peak).
#include "host_common.hpp"

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace tac {
namespace {

typedef double2 cd;

constexpr int D_THREADS = 256;

struct GeomD {
    const double* wave;
    const double* window;
    long long row_stride, rows, n_frames;
    int length, n_fft, win_length, win_offset, hop, center_pad, pad_mode, onesided;
    double scale;
};

// mode 0: complex rows [F][2]; 1: |X|^power rows [F] (+ dB when db)
struct EpiD {
    int mode, db;
    double power, amin, log10_ref;
};

__device__ __forceinline__ cd cmuld(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cd caddd(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd csubd(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }

__device__ __forceinline__ int padded_index_d(int i, int L, int mode, bool* zero) {   // F.pad semantics, as fft_core.hpp
    const int refl = i < 0 ? -i : (i >= L ? 2 * (L - 1) - i : i);
    const int clmp = i < 0 ? 0 : (i >= L ? L - 1 : i);
    const int circ = i < 0 ? i + L : (i >= L ? i - L : i);
    int j = mode == PAD_REFLECT ? refl : (mode == PAD_CIRCULAR ? circ : clmp);
    j = j < 0 ? 0 : (j >= L ? L - 1 : j);
    *zero = (mode == PAD_CONSTANT) && (i != clmp);
    return j;
}

__device__ __forceinline__ double windowed_sample(const GeomD& g, const double* __restrict__ row, int s0, int n) {
    const int wn = n - g.win_offset;
    if (wn < 0 || wn >= g.win_length) return 0.0;
    bool zero;
    const int j = padded_index_d(s0 + n, g.length, g.pad_mode, &zero);
    return zero ? 0.0 : row[j] * g.window[wn];
}

__device__ __forceinline__ double norm_pow(cd v, double power) {
    const double p2 = v.x * v.x + v.y * v.y;
    if (power == 2.0) return p2;
    const double m = sqrt(p2);
    return power == 1.0 ? m : pow(m, power);
}

__device__ __forceinline__ double to_db(double x, double amin, double log10_ref) {
    const double sq = x * x;
    return 10.0 * (log10(sq < amin ? amin : sq) - log10_ref);                // NaN stays NaN: (NaN < amin) is false
}

__device__ __forceinline__ void emit(const GeomD& g, const EpiD& ep, double* __restrict__ out, long long unit, int F, int k, cd X) {
    X.x *= g.scale;
    X.y *= g.scale;
    if (ep.mode == 0) {
        reinterpret_cast<cd*>(out)[unit * F + k] = X;
    } else {
        const double m = norm_pow(X, ep.power);
        out[unit * F + k] = ep.db ? to_db(m, ep.amin, ep.log10_ref) : m;
    }
}

// radices of the N/2-point transform: 4s, then a 2, then 3s, then 5s (plan_f64) — so the counter of a radix-4 / radix-2
// pass is a power of two
struct PlanD {
    int n, group;                                                             // passes; frames transformed side by side
    int r[14];
};

// one radix-3 / radix-5 Stockham pass over the workgroup's frames: the small transform's roots
// W_R^(t u) = W_N^((N / R)(t u mod R)) come from the table
template <int R, class Twiddle>
__device__ __forceinline__ void pass_small_radix(const cd* __restrict__ src, cd* __restrict__ dst, int M, int N, int ns, int frames,
                                                 int tid, Twiddle wn) {
    const int cnt = M / R, step = cnt / ns, unit_r = N / R;
    for (int jj = tid; jj < frames * cnt; jj += D_THREADS) {
        const int f = frames == 1 ? 0 : jj / cnt, j = jj - f * cnt, base = f * M;
        const int k = j % ns, q = k * step;
        cd v[R];
        v[0] = src[base + j];
#pragma unroll
        for (int t = 1; t < R; ++t) v[t] = cmuld(src[base + j + t * cnt], wn(2 * t * q));
        const int j0 = base + (j - k) * R + k;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            cd acc = v[0];
#pragma unroll
            for (int t = 1; t < R; ++t) acc = caddd(acc, cmuld(v[t], wn(unit_r * ((t * u) % R))));
            dst[j0 + u * ns] = acc;
        }
    }
}

// Even fft_length N = 2M <= 8192 with a 5-smooth M; `group` consecutive frames per workgroup pass (small M: keeps the 256
// threads busy).  LDS: [M + 1 twiddles when TW_LDS][group M][group M] complex doubles.
template <bool TW_LDS>
__global__ void __launch_bounds__(D_THREADS) stft_f64_kernel(GeomD g, const cd* __restrict__ tw, EpiD ep, PlanD plan,
                                                             double* __restrict__ out) {
    extern __shared__ double2 smem_d[];
    const int N = g.n_fft, M = N >> 1, tid = threadIdx.x, G = plan.group;
    cd* const wl = smem_d;
    cd* bufa = smem_d + (TW_LDS ? M + 1 : 0);
    cd* bufb = bufa + G * M;
    if (TW_LDS) {
        for (int i = tid; i <= M; i += D_THREADS) wl[i] = tw[i];
    }
    auto wn = [&](int j) -> cd {                                             // exp(-2 pi i j / N), 0 <= j < N
        const int r = j >= M ? j - M : j;
        const cd v = TW_LDS ? wl[r] : tw[r];
        return j >= M ? make_double2(-v.x, -v.y) : v;
    };
    const int F = g.onesided ? M + 1 : N;
    const long long units = g.rows * g.n_frames;
    for (long long first = (long long)blockIdx.x * G; first < units; first += (long long)gridDim.x * G) {
        const int frames = (int)(units - first < G ? units - first : G);
        __syncthreads();                                                      // the previous group's split has read its buffer
        for (int ii = tid; ii < frames * M; ii += D_THREADS) {
            const int f = frames == 1 ? 0 : ii / M, i = ii - f * M;
            const long long unit = first + f, row = unit / g.n_frames;
            const int frame = (int)(unit - row * g.n_frames);
            const double* __restrict__ rp = g.wave + row * g.row_stride;
            const int s0 = frame * g.hop - g.center_pad;
            bufa[ii] = make_double2(windowed_sample(g, rp, s0, 2 * i), windowed_sample(g, rp, s0, 2 * i + 1));
        }
        __syncthreads();
        cd* src = bufa;
        cd* dst = bufb;
        int ns = 1;
        for (int p = 0; p < plan.n; ++p) {                                    // Stockham passes, ns = product of the radices so far
            const int r = plan.r[p], cnt = M / r, step = cnt / ns;            // pass twiddle W_M^(t k step) = W_N^(2 t k step)
            if (r == 4) {
                for (int jj = tid; jj < frames * cnt; jj += D_THREADS) {
                    const int f = frames == 1 ? 0 : jj / cnt, j = jj - f * cnt, base = f * M;
                    const int k = j & (ns - 1), q = k * step;
                    const cd v0 = src[base + j], v1 = cmuld(src[base + j + cnt], wn(2 * q)),
                             v2 = cmuld(src[base + j + 2 * cnt], wn(4 * q)), v3 = cmuld(src[base + j + 3 * cnt], wn(6 * q));
                    const cd s0c = caddd(v0, v2), s1c = csubd(v0, v2), s2c = caddd(v1, v3), s3c = csubd(v1, v3);
                    const int j0 = base + ((j - k) << 2) + k;
                    dst[j0] = caddd(s0c, s2c);
                    dst[j0 + ns] = make_double2(s1c.x + s3c.y, s1c.y - s3c.x);    // s1 - i s3
                    dst[j0 + 2 * ns] = csubd(s0c, s2c);
                    dst[j0 + 3 * ns] = make_double2(s1c.x - s3c.y, s1c.y + s3c.x);
                }
            } else if (r == 2) {
                for (int jj = tid; jj < frames * cnt; jj += D_THREADS) {
                    const int f = frames == 1 ? 0 : jj / cnt, j = jj - f * cnt, base = f * M;
                    const int k = j & (ns - 1);
                    const cd v0 = src[base + j], v1 = cmuld(src[base + j + cnt], wn(2 * k * step));
                    const int j0 = base + ((j - k) << 1) + k;
                    dst[j0] = caddd(v0, v1);
                    dst[j0 + ns] = csubd(v0, v1);
                }
            } else if (r == 3) {
                pass_small_radix<3>(src, dst, M, N, ns, frames, tid, wn);
            } else {
                pass_small_radix<5>(src, dst, M, N, ns, frames, tid, wn);
            }
            __syncthreads();
            cd* t = src; src = dst; dst = t;
            ns *= r;
        }
        // real-input split: X[k] = (Z[k] + conj Z[M-k]) / 2 - i W_N^k (Z[k] - conj Z[M-k]) / 2, k = 0..M
        for (int kk = tid; kk < frames * (M + 1); kk += D_THREADS) {
            const int f = frames == 1 ? 0 : kk / (M + 1), k = kk - f * (M + 1);
            const cd* __restrict__ z = src + f * M;
            const long long unit = first + f;
            const cd a = z[k == M ? 0 : k], braw = z[k == 0 ? 0 : M - k];
            const cd b = make_double2(braw.x, -braw.y);
            const cd e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y)), o = make_double2(0.5 * (a.x - b.x), 0.5 * (a.y - b.y));
            const cd wo = cmuld(k == M ? make_double2(-1.0, 0.0) : wn(k), o);
            const cd X = make_double2(e.x + wo.y, e.y - wo.x);               // e - i (w o)
            emit(g, ep, out, unit, F, k, X);
            if (!g.onesided && k > 0 && k < M) emit(g, ep, out, unit, F, N - k, make_double2(X.x, -X.y));
        }
    }
}

// Any other fft_length <= 4096 (odd, or N/2 with a prime factor above 5): the windowed frame (N doubles) and all N twiddles in LDS, one output bin per thread pass.
__global__ void __launch_bounds__(D_THREADS) stft_f64_direct_kernel(GeomD g, const cd* __restrict__ tw, EpiD ep, double* __restrict__ out) {
    extern __shared__ double2 smem_d[];
    const int N = g.n_fft, tid = threadIdx.x;
    cd* const wl = smem_d;
    double* const x = reinterpret_cast<double*>(smem_d + N);
    for (int i = tid; i < N; i += D_THREADS) wl[i] = tw[i];
    const int half = N / 2;
    const int F = g.onesided ? half + 1 : N;
    const long long units = g.rows * g.n_frames;
    for (long long unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const long long row = unit / g.n_frames;
        const int frame = (int)(unit - row * g.n_frames);
        const double* __restrict__ rp = g.wave + row * g.row_stride;
        const int s0 = frame * g.hop - g.center_pad;
        __syncthreads();
        for (int i = tid; i < N; i += D_THREADS) x[i] = windowed_sample(g, rp, s0, i);
        __syncthreads();
        for (int k = tid; k <= half; k += D_THREADS) {
            double re = 0.0, im = 0.0;
            int idx = 0;
            for (int n = 0; n < N; ++n) {
                const cd w = wl[idx];
                re = fma(x[n], w.x, re);
                im = fma(x[n], w.y, im);
                idx += k;
                idx -= idx >= N ? N : 0;
            }
            const cd X = make_double2(re, im);
            emit(g, ep, out, unit, F, k, X);
            if (!g.onesided && k > 0 && 2 * k != N) emit(g, ep, out, unit, F, N - k, make_double2(re, -im));
        }
    }
}

int twiddles_f64(int n_fft, const cd** out) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, const cd*> cache;
    int dev = 0;
    TAC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(n_fft, dev);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = it->second;
        return TAC_OK;
    }
    std::vector<cd> host((size_t)n_fft + 1);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k <= n_fft; ++k) {
        // exact octant values (k a multiple of N/4) stay exact; everything else is rounded once from long double
        const long double a = -two_pi * (long double)k / (long double)n_fft;
        double c = (double)cosl(a), s = (double)sinl(a);
        if ((4LL * k) % n_fft == 0) {
            const int quad = (int)((4LL * k) / n_fft) & 3;
            c = quad == 0 ? 1.0 : (quad == 2 ? -1.0 : 0.0);
            s = quad == 1 ? -1.0 : (quad == 3 ? 1.0 : 0.0);
        }
        host[k] = make_double2(c, s);
    }
    cd* dptr = nullptr;
    TAC_HIP(hipMalloc((void**)&dptr, host.size() * sizeof(cd)));
    TAC_HIP(hipMemcpy(dptr, host.data(), host.size() * sizeof(cd), hipMemcpyHostToDevice));
    cache[key] = dptr;
    *out = dptr;
    return TAC_OK;
}

// N even, N/2 = 4^a 2^b 3^c 5^d: the Stockham kernel's pass list (false: the direct kernel's case)
bool plan_f64(int n_fft, PlanD* plan) {
    plan->n = 0;
    plan->group = 1;
    if (n_fft < 4 || (n_fft & 1) || n_fft > 8192) return false;
    int m = n_fft / 2;
    const int radices[4] = {4, 2, 3, 5};
    for (int r : radices)
        while (m % r == 0 && plan->n < 14) {
            plan->r[plan->n++] = r;
            m /= r;
        }
    const int per = 1024 / (n_fft / 2);                                      // ~4 radix-4 butterflies per thread and pass
    plan->group = per < 1 ? 1 : (per > 16 ? 16 : per);
    return m == 1 && plan->n > 0;
}

int geometry_f64(const double* wave, const double* window, const tac_stft_desc* d, GeomD* g) {
    if (!wave || !window || !d) return TAC_E_INVALID;
    if (d->rows <= 0 || d->length <= 0 || d->hop <= 0 || d->n_fft <= 0) return TAC_E_INVALID;
    if (d->win_length <= 0 || d->win_length > d->n_fft) return TAC_E_INVALID;
    if (d->pad_mode < TAC_PAD_CONSTANT || d->pad_mode > TAC_PAD_CIRCULAR) return TAC_E_INVALID;
    if (d->row_stride < d->length) return TAC_E_INVALID;
    if (d->length >= 0x7fffffffLL - 2 * (int64_t)d->n_fft) return TAC_E_UNSUPPORTED;
    PlanD plan;
    if (!plan_f64(d->n_fft, &plan) && d->n_fft > 4096) return TAC_E_UNSUPPORTED;
    const int pad = d->center ? d->n_fft / 2 : 0;
    if (pad > 0) {
        if (d->pad_mode == TAC_PAD_REFLECT && pad >= d->length) return TAC_E_SHORT_INPUT;
        if (d->pad_mode == TAC_PAD_CIRCULAR && pad > d->length) return TAC_E_SHORT_INPUT;
    }
    const int64_t T = tac_num_frames(d->length, d->n_fft, d->hop, d->center);
    if (T <= 0) return TAC_E_SHORT_INPUT;
    g->wave = wave;
    g->window = window;
    g->row_stride = d->row_stride;
    g->rows = d->rows;
    g->n_frames = T;
    g->length = (int)d->length;
    g->n_fft = d->n_fft;
    g->win_length = d->win_length;
    g->win_offset = (d->n_fft - d->win_length) / 2;
    g->hop = d->hop;
    g->center_pad = pad;
    g->pad_mode = d->pad_mode;
    g->onesided = d->onesided ? 1 : 0;
    g->scale = d->normalized ? 1.0 / std::sqrt((double)d->n_fft) : 1.0;
    return TAC_OK;
}

int launch_stft_f64(const double* wave, const double* window, const tac_stft_desc* d, const EpiD& ep, double* out, hipStream_t stream) {
    if (!out) return TAC_E_INVALID;
    GeomD g;
    int rc = geometry_f64(wave, window, d, &g);
    if (rc != TAC_OK) return rc;
    const cd* tw = nullptr;
    rc = twiddles_f64(d->n_fft, &tw);
    if (rc != TAC_OK) return rc;
    const long long units = g.rows * g.n_frames;
    long long blocks = units;
    const long long cap = (long long)device_cu_count() * 8;
    if (blocks > cap) blocks = cap;
    const int N = d->n_fft, M = N / 2;
    PlanD plan;
    if (plan_f64(N, &plan)) {
        const bool tw_lds = N <= 4096;
        const size_t lds = ((size_t)(tw_lds ? M + 1 : 0) + 2 * (size_t)M * plan.group) * sizeof(cd);
        blocks = (units + plan.group - 1) / plan.group;
        if (blocks > cap) blocks = cap;
        if (tw_lds) {
            TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(stft_f64_kernel<true>), (int)lds));
            hipLaunchKernelGGL(stft_f64_kernel<true>, dim3((unsigned)blocks), dim3(D_THREADS), lds, stream, g, tw, ep, plan, out);
        } else {
            TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(stft_f64_kernel<false>), (int)lds));
            hipLaunchKernelGGL(stft_f64_kernel<false>, dim3((unsigned)blocks), dim3(D_THREADS), lds, stream, g, tw, ep, plan, out);
        }
    } else {
        const size_t lds = (size_t)N * sizeof(cd) + (size_t)N * sizeof(double);
        TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(stft_f64_direct_kernel), (int)lds));
        hipLaunchKernelGGL(stft_f64_direct_kernel, dim3((unsigned)blocks), dim3(D_THREADS), lds, stream, g, tw, ep, out);
    }
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// ------------------------------------------------------------------ filterbank contraction, float64
// out[r][t][m] = sum_f spec(r, f, t) fb[f][m] (+ dB): 64 frames x 64 bands per workgroup, 4 x 4 outputs per thread,
// 16-bin slabs through LDS.  spec is addressed through (row, freq, frame) strides like the float32 GEMM entry.
constexpr int FB_TM = 64, FB_TN = 64, FB_TK = 16;

__global__ void __launch_bounds__(256) fb_f64_kernel(const double* __restrict__ spec, long long stride_r, long long stride_f,
                                                     long long stride_t, int n_freqs, int n_frames,
                                                     const double* __restrict__ fb, int n_mels, EpiD ep, double* __restrict__ out) {
    __shared__ double sa[FB_TK][FB_TM + 1];
    __shared__ double sb[FB_TK][FB_TN + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const long long row = blockIdx.z;
    const int t0 = blockIdx.x * FB_TM, m0 = blockIdx.y * FB_TN;
    const double* __restrict__ sp = spec + row * stride_r;
    double acc[4][4] = {};
    const bool frames_fast = stride_t == 1;                                   // which index runs along the 64 lanes of a load
    for (int f0 = 0; f0 < n_freqs; f0 += FB_TK) {
        for (int i = tid; i < FB_TK * FB_TM; i += 256) {
            const int kk = frames_fast ? i / FB_TM : i % FB_TK, tt = frames_fast ? i % FB_TM : i / FB_TK;
            const int f = f0 + kk, t = t0 + tt;
            sa[kk][tt] = (f < n_freqs && t < n_frames) ? sp[(long long)f * stride_f + (long long)t * stride_t] : 0.0;
        }
        for (int i = tid; i < FB_TK * FB_TN; i += 256) {
            const int kk = i / FB_TN, mm = i % FB_TN;
            const int f = f0 + kk, m = m0 + mm;
            sb[kk][mm] = (f < n_freqs && m < n_mels) ? fb[(long long)f * n_mels + m] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < FB_TK; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sa[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sb[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 16 * i;
        if (t >= n_frames) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + tx + 16 * j;
            if (m < n_mels) out[(row * n_frames + t) * n_mels + m] = ep.db ? to_db(acc[i][j], ep.amin, ep.log10_ref) : acc[i][j];
        }
    }
}

// ------------------------------------------------------------------ elementwise, float64
enum { EW_NORM = 0, EW_ANGLE = 1, EW_DB = 2, EW_UNDB = 3 };

template <int OP>
__global__ void __launch_bounds__(256) ew_f64_kernel(const double* __restrict__ x, long long n, double p0, double p1,
                                                     double* __restrict__ out, double* __restrict__ out2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (OP == EW_NORM) {                                                  // p0 = power; out2 (optional) = angle
            const cd v = reinterpret_cast<const cd*>(x)[i];
            if (out) out[i] = norm_pow(v, p0);
            if (out2) out2[i] = atan2(v.y, v.x);
        } else if (OP == EW_DB) {                                             // p0 = amin, p1 = log10(ref)
            out[i] = to_db(x[i], p0, p1);
        } else if (OP == EW_UNDB) {                                           // p1 = log10(ref): sqrt(10^(x / 10 + log10 ref))
            out[i] = sqrt(pow(10.0, x[i] / 10.0 + p1));
        }
    }
}

template <int OP>
int launch_ew(const double* x, int64_t n, double p0, double p1, double* out, double* out2, void* stream) {
    if (!x || n < 0 || (!out && !out2)) return TAC_E_INVALID;
    if (n == 0) return TAC_OK;
    long long blocks = (n + 255) / 256;
    const long long cap = (long long)device_cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ew_f64_kernel<OP>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, p0, p1, out, out2);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // namespace
}  // namespace tac

extern "C" {

int tac_stft_f64(const double* wave, const double* window, const tac_stft_desc* d, double* out, void* stream) {
    using namespace tac;
    return launch_stft_f64(wave, window, d, EpiD{0, 0, 1.0, 0.0, 0.0}, out, (hipStream_t)stream);
}

int tac_spectrogram_f64(const double* wave, const double* window, const tac_stft_desc* d, double power, int db, double db_ref,
                        double db_amin, double* out, void* stream) {
    using namespace tac;
    if (db && !(db_ref > 0.0)) return TAC_E_INVALID;
    return launch_stft_f64(wave, window, d, EpiD{1, db ? 1 : 0, power, db_amin, db ? std::log10(db_ref) : 0.0}, out,
                           (hipStream_t)stream);
}

int tac_apply_filterbank_f64(const double* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                             int64_t stride_f, int64_t stride_t, const double* fb, int32_t n_mels, int db, double db_ref,
                             double db_amin, double* out, void* stream) {
    using namespace tac;
    if (!spec || !fb || !out || rows <= 0 || n_freqs <= 0 || n_frames <= 0 || n_mels <= 0) return TAC_E_INVALID;
    if (db && !(db_ref > 0.0)) return TAC_E_INVALID;
    if (n_frames >= 0x7fffffffLL || rows > 65535) return TAC_E_UNSUPPORTED;
    const dim3 grid((unsigned)((n_frames + FB_TM - 1) / FB_TM), (unsigned)((n_mels + FB_TN - 1) / FB_TN), (unsigned)rows);
    if (grid.y > 65535) return TAC_E_UNSUPPORTED;
    hipLaunchKernelGGL(fb_f64_kernel, grid, dim3(256), 0, (hipStream_t)stream, spec, (long long)stride_r, (long long)stride_f,
                       (long long)stride_t, (int)n_freqs, (int)n_frames, fb, (int)n_mels,
                       EpiD{1, db ? 1 : 0, 1.0, db_amin, db ? std::log10(db_ref) : 0.0}, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_magphase_f64(const double* z, int64_t n, double power, double* mag, double* phase, void* stream) {
    return tac::launch_ew<tac::EW_NORM>(z, n, power, 0.0, mag, phase, stream);
}

int tac_amplitude_to_db_f64(const double* x, int64_t n, double ref, double amin, double* out, void* stream) {
    if (!(ref > 0.0) || !out) return TAC_E_INVALID;
    return tac::launch_ew<tac::EW_DB>(x, n, amin, std::log10(ref), out, nullptr, stream);
}

int tac_db_to_amplitude_f64(const double* x, int64_t n, double ref, double* out, void* stream) {
    if (!(ref > 0.0) || !out) return TAC_E_INVALID;
    return tac::launch_ew<tac::EW_UNDB>(x, n, 0.0, std::log10(ref), out, nullptr, stream);
}

}  // extern "C"
