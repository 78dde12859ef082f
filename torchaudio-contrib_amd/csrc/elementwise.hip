// elementwise.hip — HBM-bound streaming kernels of the path: complex_norm, amplitude_to_db,
// db_to_amplitude, mu-law encode / decode (functional.py:116-128, 277-314, 317-354).
// 16 B per lane per access where alignment allows, grid-stride over 256 CUs x 8 blocks.
#include <cstdlib>

#include "host_common.hpp"
#include "exact_math.hpp"

namespace tac {

constexpr int EW_THREADS = 256;
// nontemporal 16-byte store of a float4 (through a native vector type: the builtin takes only those).  Round 5, same box, alternating
// processes (profiles/r05/ab/batch21 ... 23): for the 1 : 1 maps whose output nobody re-reads inside the launch — the unary map
// (dB both ways) 68 -> 73 % of the HBM peak, mu-law decode 70 -> 72 % — it pays; for complex_norm / magphase it does not (+-1 %),
// for the mu-law encoder's int64 stores it costs 40 %, and nontemporal LOADS cost 3 - 9 % everywhere: those stay plain.
typedef float ew_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ew_store_once(float4* p, float4 v) {
    const ew_f4 u = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(u, reinterpret_cast<ew_f4*>(p));
}

// Blocks per CU of the grid-stride kernels.  On this chip a pure fill runs 5.6 TB/s from 2 blocks of 256 threads per CU and
// 4.4-4.6 from 4-16, a pure read needs >= 4 (tools/ubench/hbm_rate.hip) — so every kernel carries the count its read : write
// mix measured best with at cfg-2 / cfg-5 sizes (profiles/r04/ab/batch11_ew_blocks.txt): complex_norm (2 : 1) 2 blocks =
// 0.167 vs 0.189 ms with 8; magphase (1 : 1, two outputs) 3 = 0.247 vs 0.267; mu-law encode (1 : 2) 3 = 0.309 vs 0.345; the dB
// pair and mu-law decode 8.  TAC_EW_BLOCKS_PER_CU overrides all of them (A/B runs).
constexpr int EW_DEFAULT = 8, EW_COMPLEX_NORM = 2, EW_MAGPHASE = 3, EW_MULAW_ENCODE = 3;
static inline int ew_blocks_override() {
    static const int v = [] { const char* e = getenv("TAC_EW_BLOCKS_PER_CU"); const int n = e ? atoi(e) : 0; return n > 0 ? n : 0; }();
    return v;
}
static inline unsigned ew_blocks(long long work_items, int per_cu = EW_DEFAULT) {
    long long cap = (long long)device_cu_count() * (ew_blocks_override() ? ew_blocks_override() : per_cu);
    long long want = (work_items + EW_THREADS - 1) / EW_THREADS;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------- complex_norm
__device__ __forceinline__ float norm_pow(float re, float im, float power) {
    return cpow_mag(mkc(re, im), power);
}

template <bool VEC>
__global__ void __launch_bounds__(EW_THREADS) complex_norm_kernel(const float* __restrict__ x, long long n, float power,
                                                                  float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const long long n4 = n / 4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (long long j = i; j < n4; j += stride) {
            float4 a = x4[2 * j], b = x4[2 * j + 1];
            o4[j] = make_float4(norm_pow(a.x, a.y, power), norm_pow(a.z, a.w, power), norm_pow(b.x, b.y, power),
                                norm_pow(b.z, b.w, power));
        }
        for (long long j = n4 * 4 + i; j < n; j += stride) out[j] = norm_pow(x[2 * j], x[2 * j + 1], power);
    } else {
        for (long long j = i; j < n; j += stride) out[j] = norm_pow(x[2 * j], x[2 * j + 1], power);
    }
}

// ---------------------------------------------------------------- angle / magphase (functional.py:187-201)
// one pass over the complex pairs: phase = atan2(im, re), optionally also |z|^power (dual output)
template <bool VEC, bool WITH_MAG>
__global__ void __launch_bounds__(EW_THREADS) magphase_kernel(const float* __restrict__ x, long long n, float power,
                                                              float* __restrict__ mag, float* __restrict__ phase) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const long long n4 = n / 4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* p4 = reinterpret_cast<float4*>(phase);
        float4* m4 = reinterpret_cast<float4*>(mag);
        for (long long j = i; j < n4; j += stride) {
            const float4 a = x4[2 * j], b = x4[2 * j + 1];
            p4[j] = make_float4(atan2f(a.y, a.x), atan2f(a.w, a.z), atan2f(b.y, b.x), atan2f(b.w, b.z));
            if constexpr (WITH_MAG)
                m4[j] = make_float4(norm_pow(a.x, a.y, power), norm_pow(a.z, a.w, power), norm_pow(b.x, b.y, power),
                                    norm_pow(b.z, b.w, power));
        }
        for (long long j = n4 * 4 + i; j < n; j += stride) {
            phase[j] = atan2f(x[2 * j + 1], x[2 * j]);
            if constexpr (WITH_MAG) mag[j] = norm_pow(x[2 * j], x[2 * j + 1], power);
        }
    } else {
        for (long long j = i; j < n; j += stride) {
            phase[j] = atan2f(x[2 * j + 1], x[2 * j]);
            if constexpr (WITH_MAG) mag[j] = norm_pow(x[2 * j], x[2 * j + 1], power);
        }
    }
}

// ---------------------------------------------------------------- generic unary map
struct AmpToDb {
    float amin, log10_ref;
    __device__ __forceinline__ float operator()(float v) const { return amp_to_db(v, amin, log10_ref); }
};
struct DbToAmp {
    float log10_ref;
    // (10^(x/10 + log10 ref))^0.5
    // = 2^((x/10 + log10 ref) * log2(10) / 2): one exp2 instead of powf + sqrtf (the kernel was ALU-bound at 2 TB/s)
    __device__ __forceinline__ float operator()(float v) const { return exp2f((v / 10.0f + log10_ref) * 1.6609640474436813f); }
};

template <bool VEC, class Op>
__global__ void __launch_bounds__(EW_THREADS) unary_kernel(const float* __restrict__ x, long long n, Op op,
                                                           float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const long long n4 = n / 4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (long long j = i; j < n4; j += stride) {
            float4 a = x4[j];
            ew_store_once(o4 + j, make_float4(op(a.x), op(a.y), op(a.z), op(a.w)));
        }
        for (long long j = n4 * 4 + i; j < n; j += stride) out[j] = op(x[j]);
    } else {
        for (long long j = i; j < n; j += stride) out[j] = op(x[j]);
    }
}

// ---------------------------------------------------------------- mu-law
// closed form in the reference's op order (functional.py:331-334), fp32, every operation individually rounded
__device__ __forceinline__ long long mulaw_formula(float x, float mu, float log1p_mu) {
#pragma clang fp contract(off)
    float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : x);     // torch.sign: 0 -> 0, NaN -> NaN
    float comp = sgn * exact_log1pf(mu * fabsf(x)) / log1p_mu;
    float q = (comp + 1.0f) / 2.0f * mu + 0.5f;
    if (!(fabsf(q) < 9.2233720e18f)) return (long long)0x8000000000000000ULL;   // x86 cvttss2si "indefinite"
    return (long long)q;                                          // trunc toward zero == .long()
}

constexpr int MULAW_MAX_THR = 1024;

// float64: the same op order in double
__global__ void __launch_bounds__(EW_THREADS)
mulaw_encode_f64_kernel(const double* __restrict__ x, long long n, double mu, long long* __restrict__ out) {
#pragma clang fp contract(off)
    const double l1p = log1p(mu);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const double v = x[j];
        const double sgn = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : v);
        const double q = (sgn * log1p(mu * fabs(v)) / l1p + 1.0) / 2.0 * mu + 0.5;
        out[j] = (fabs(q) < 9.2233720368547758e18) ? (long long)q : (long long)0x8000000000000000ULL;
    }
}

template <class CODE>
__global__ void __launch_bounds__(EW_THREADS)
mulaw_decode_f64_kernel(const CODE* __restrict__ codes, long long n, double mu, double* __restrict__ out) {
#pragma clang fp contract(off)
    const double l1p = log1p(mu);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const double y = ((double)codes[j] / mu) * 2.0 - 1.0;
        const double sgn = (y > 0.0) ? 1.0 : ((y < 0.0) ? -1.0 : y);
        out[j] = sgn * (exp(fabs(y) * l1p) - 1.0) / mu;
    }
}

template <bool VEC>
__global__ void __launch_bounds__(EW_THREADS)
mulaw_encode_kernel(const float* __restrict__ x, long long n, float mu, float log1p_mu, const int* __restrict__ thr,
                    int n_pos, int n_neg, int zero_code, long long* __restrict__ out) {
    __shared__ int s_thr[MULAW_MAX_THR];
    const bool use_thr = thr != nullptr;
    if (use_thr) {
        for (int i = threadIdx.x; i < n_pos + n_neg; i += blockDim.x) s_thr[i] = thr[i];
        __syncthreads();
    }
    // |x| <= 1 with a threshold table: the code is the number of thresholds <= |x|'s bit pattern.  A cheap
    // estimate (hardware log2, within +-1 code) picks the starting index and one branch-free compare step in
    // each direction make it exact; the precise closed form is only evaluated for out-of-range inputs.
    const float est_scale = 0.5f * mu / (log1p_mu * 1.4426950408889634f);         // codes per log2 unit
    auto encode = [&](float v) -> long long {
        const unsigned raw = __float_as_uint(v);
        const int bits = (int)(raw & 0x7fffffffu);
        if (!use_thr || bits > 0x3f800000) return mulaw_formula(v, mu, log1p_mu);   // no table, |x| > 1 or NaN
        const bool neg = (raw >> 31) != 0;
        const int* tb = neg ? s_thr + n_pos : s_thr;
        const int nt = neg ? n_neg : n_pos;
        int cnt = (int)(__log2f(1.0f + mu * __uint_as_float((unsigned)bits)) * est_scale + 0.5f);
        cnt = cnt < 0 ? 0 : (cnt > nt ? nt : cnt);
#pragma unroll
        for (int r = 0; r < 1; ++r) {       // one step each way: exhaustively verified on device (tools/check_mulaw_exhaustive.py)
            cnt += (cnt < nt && tb[cnt < nt ? cnt : nt - 1] <= bits) ? 1 : 0;
            cnt -= (cnt > 0 && tb[cnt > 0 ? cnt - 1 : 0] > bits) ? 1 : 0;
        }
        return neg ? (long long)(zero_code - cnt) : (long long)(zero_code + cnt);
    };
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const long long n4 = n / 4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        longlong2* o2 = reinterpret_cast<longlong2*>(out);
        for (long long j = i; j < n4; j += stride) {
            float4 a = x4[j];
            o2[2 * j] = make_longlong2(encode(a.x), encode(a.y));
            o2[2 * j + 1] = make_longlong2(encode(a.z), encode(a.w));
        }
        for (long long j = n4 * 4 + i; j < n; j += stride) out[j] = encode(x[j]);
    } else {
        for (long long j = i; j < n; j += stride) out[j] = encode(x[j]);
    }
}

// closed form of functional.py:352-353
__device__ __forceinline__ float mulaw_expand(float code, float mu, float log1p_mu) {
    float y = (code / mu) * 2.0f - 1.0f;
    float sgn = (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : y);
    return sgn * (expf(fabsf(y) * log1p_mu) - 1.0f) / mu;
}

template <bool VEC>
__global__ void __launch_bounds__(EW_THREADS)
mulaw_decode_i64_kernel(const long long* __restrict__ codes, long long n, int nq, float mu, float log1p_mu,
                        const float* __restrict__ lut, float* __restrict__ out) {
    auto decode = [&](long long c) -> float {
        if (lut != nullptr && c >= 0 && c < nq) return lut[c];
        return mulaw_expand((float)c, mu, log1p_mu);
    };
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const long long n4 = n / 4;
        const longlong2* c2 = reinterpret_cast<const longlong2*>(codes);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (long long j = i; j < n4; j += stride) {
            longlong2 a = c2[2 * j], b = c2[2 * j + 1];
            ew_store_once(o4 + j, make_float4(decode(a.x), decode(a.y), decode(b.x), decode(b.y)));
        }
        for (long long j = n4 * 4 + i; j < n; j += stride) out[j] = decode(codes[j]);
    } else {
        for (long long j = i; j < n; j += stride) out[j] = decode(codes[j]);
    }
}

// float-valued codes: an integral code inside the table takes the table entry (the reference bit-compares exactly
// this input, tests/test_functional.py:182-193: `waveform_mu.float()`), everything else the closed form
struct MulawExpandOp {
    float mu, log1p_mu;
    const float* lut;
    int nq;
    __device__ __forceinline__ float operator()(float c) const {
        if (lut != nullptr && c >= 0.0f && c < (float)nq) {
            const int i = (int)c;
            if ((float)i == c) return lut[i];
        }
        return mulaw_expand(c, mu, log1p_mu);
    }
};

template <class Op>
static int launch_unary(const float* x, int64_t n, Op op, float* out, void* stream) {
    if (n == 0) return TAC_OK;
    if (!x || !out || n < 0) return TAC_E_INVALID;
    const bool vec = aligned16(x) && aligned16(out);
    const unsigned blocks = ew_blocks(vec ? (n + 3) / 4 : n);
    if (vec) hipLaunchKernelGGL((unary_kernel<true, Op>), dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, (long long)n, op, out);
    else hipLaunchKernelGGL((unary_kernel<false, Op>), dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, (long long)n, op, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

__global__ void __launch_bounds__(EW_THREADS) pcm16_kernel(const short* __restrict__ x, long long n, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) out[j] = (float)x[j] * (1.0f / 32768.0f);
}

// Gradient of a TWO-sided output folded onto the one-sided bins the gradient kernels take: for a real signal
// X[N - k] = conj(X[k]), so d/dx through bin N - k equals d/dx through bin k with the imaginary part's gradient negated
// (width 2: (re, im) pairs of the complex stft; width 1: |X|^p values, which simply add).  Bins 0 and N/2 have no twin.
__global__ void __launch_bounds__(EW_THREADS)
fold_twosided_kernel(const float* __restrict__ grad, long long n, int n_fft, int n_bins, int width, float* __restrict__ out) {
    const long long per_frame = (long long)n_bins * width;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long frame = i / per_frame;
        const int r = (int)(i - frame * per_frame);
        const int k = r / width, c = r - k * width;
        const float* src = grad + frame * (long long)n_fft * width;
        float v = src[(long long)k * width + c];
        if (k > 0 && 2 * k != n_fft) {
            const float twin = src[(long long)(n_fft - k) * width + c];
            v += (c == 1) ? -twin : twin;
        }
        out[i] = v;
    }
}

// out[i] = sum over slabs s of x[s][i], added in slab order (deterministic)
__global__ void __launch_bounds__(EW_THREADS)
sum_slabs_kernel(const float* __restrict__ x, long long n_slabs, long long slab_elems, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < slab_elems; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (long long s2 = 0; s2 < n_slabs; ++s2) acc += x[s2 * slab_elems + i];
        out[i] = acc;
    }
}

}  // namespace tac

extern "C" {

int tac_pcm16_to_f32(const int16_t* x, int64_t n, float* out, void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!x || !out || n < 0) return TAC_E_INVALID;
    hipLaunchKernelGGL(pcm16_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, (long long)n, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_complex_norm_f32(const float* x, int64_t n, float power, float* out, void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!x || !out || n < 0) return TAC_E_INVALID;
    const bool vec = aligned16(x) && aligned16(out);
    const unsigned blocks = ew_blocks(vec ? (n + 3) / 4 : n, EW_COMPLEX_NORM);
    if (vec) hipLaunchKernelGGL(complex_norm_kernel<true>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, (long long)n, power, out);
    else hipLaunchKernelGGL(complex_norm_kernel<false>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, (long long)n, power, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_magphase_f32(const float* x, int64_t n, float power, float* mag, float* phase, void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!x || !phase || n < 0) return TAC_E_INVALID;
    const bool vec = aligned16(x) && aligned16(phase) && (!mag || aligned16(mag));
    const unsigned blocks = ew_blocks(vec ? (n + 3) / 4 : n, EW_MAGPHASE);
    const hipStream_t s = (hipStream_t)stream;
    if (mag) {
        if (vec) hipLaunchKernelGGL((magphase_kernel<true, true>), dim3(blocks), dim3(EW_THREADS), 0, s, x, (long long)n, power, mag, phase);
        else hipLaunchKernelGGL((magphase_kernel<false, true>), dim3(blocks), dim3(EW_THREADS), 0, s, x, (long long)n, power, mag, phase);
    } else {
        if (vec) hipLaunchKernelGGL((magphase_kernel<true, false>), dim3(blocks), dim3(EW_THREADS), 0, s, x, (long long)n, power, mag, phase);
        else hipLaunchKernelGGL((magphase_kernel<false, false>), dim3(blocks), dim3(EW_THREADS), 0, s, x, (long long)n, power, mag, phase);
    }
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_amplitude_to_db_f32(const float* x, int64_t n, float ref, float amin, float* out, void* stream) {
    return tac::launch_unary(x, n, tac::AmpToDb{amin, log10f(ref)}, out, stream);
}

int tac_db_to_amplitude_f32(const float* x, int64_t n, float ref, float* out, void* stream) {
    return tac::launch_unary(x, n, tac::DbToAmp{log10f(ref)}, out, stream);
}

int tac_mulaw_encode_f32_i64(const float* x, int64_t n, int32_t n_quantize, const int32_t* thresholds, int32_t n_pos,
                             int32_t n_neg, int32_t zero_code, int64_t* out, void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!x || !out || n < 0 || n_quantize < 2) return TAC_E_INVALID;
    if (thresholds && (n_pos < 0 || n_neg < 0 || n_pos + n_neg > MULAW_MAX_THR)) return TAC_E_UNSUPPORTED;
    const float mu = (float)(n_quantize - 1);
    const float l1p = exact_log1pf(mu);
    const bool vec = aligned16(x) && aligned16(out);
    const unsigned blocks = ew_blocks(vec ? (n + 3) / 4 : n, EW_MULAW_ENCODE);
    long long* o = reinterpret_cast<long long*>(out);
    if (vec) hipLaunchKernelGGL(mulaw_encode_kernel<true>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, (long long)n, mu, l1p, thresholds, n_pos, n_neg, zero_code, o);
    else hipLaunchKernelGGL(mulaw_encode_kernel<false>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, (long long)n, mu, l1p, thresholds, n_pos, n_neg, zero_code, o);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_mulaw_decode_i64_f32(const int64_t* codes, int64_t n, int32_t n_quantize, const float* lut, float* out,
                             void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!codes || !out || n < 0 || n_quantize < 2) return TAC_E_INVALID;
    const float mu = (float)(n_quantize - 1);
    const float l1p = exact_log1pf(mu);
    const bool vec = aligned16(codes) && aligned16(out);
    const unsigned blocks = ew_blocks(vec ? (n + 3) / 4 : n);
    const long long* c = reinterpret_cast<const long long*>(codes);
    if (vec) hipLaunchKernelGGL(mulaw_decode_i64_kernel<true>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, c, (long long)n, n_quantize, mu, l1p, lut, out);
    else hipLaunchKernelGGL(mulaw_decode_i64_kernel<false>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, c, (long long)n, n_quantize, mu, l1p, lut, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_mulaw_decode_f32_f32(const float* codes, int64_t n, int32_t n_quantize, const float* lut, float* out,
                             void* stream) {
    if (n_quantize < 2) return TAC_E_INVALID;
    const float mu = (float)(n_quantize - 1);
    return tac::launch_unary(codes, n, tac::MulawExpandOp{mu, tac::exact_log1pf(mu), lut, n_quantize}, out, stream);
}

// ---- float64 mu-law (round 5): the reference's formulas evaluated in double like its CPU path does for double input
//      (functional.py:329-335, 349-354).  Elementwise, 8 bytes in / 8 out; the transition between two codes sits where
//      (comp + 1) / 2 * mu + 0.5 crosses an integer — a double log1p that differs from the host's by an ulp moves a code only for
//      inputs within ~1e-16 (relative) of such a point.
int tac_mulaw_encode_f64_i64(const double* x, int64_t n, int32_t n_quantize, int64_t* out, void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!x || !out || n < 0 || n_quantize < 2) return TAC_E_INVALID;
    hipLaunchKernelGGL(mulaw_encode_f64_kernel, dim3(ew_blocks(n, EW_MULAW_ENCODE)), dim3(EW_THREADS), 0, (hipStream_t)stream, x,
                       (long long)n, (double)(n_quantize - 1), reinterpret_cast<long long*>(out));
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_mulaw_decode_f64(const void* codes, int32_t codes_are_i64, int64_t n, int32_t n_quantize, double* out, void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!codes || !out || n < 0 || n_quantize < 2) return TAC_E_INVALID;
    const double mu = (double)(n_quantize - 1);
    if (codes_are_i64)
        hipLaunchKernelGGL(mulaw_decode_f64_kernel<long long>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                           static_cast<const long long*>(codes), (long long)n, mu, out);
    else
        hipLaunchKernelGGL(mulaw_decode_f64_kernel<double>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                           static_cast<const double*>(codes), (long long)n, mu, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// ---- helpers of the general gradient routes (gradients of two-sided outputs, of the window and of the filterbank)
int tac_fold_twosided_f32(const float* grad, int64_t n_frames_total, int32_t n_fft, int32_t width, float* out,
                          void* stream) {
    using namespace tac;
    if (n_frames_total == 0) return TAC_OK;
    if (!grad || !out || n_frames_total < 0 || n_fft < 1 || (width != 1 && width != 2)) return TAC_E_INVALID;
    const int n_bins = n_fft / 2 + 1;
    const long long n = (long long)n_frames_total * n_bins * width;
    hipLaunchKernelGGL(fold_twosided_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, grad, n, n_fft,
                       n_bins, width, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_sum_slabs_f32(const float* x, int64_t n_slabs, int64_t slab_elems, float* out, void* stream) {
    using namespace tac;
    if (slab_elems == 0) return TAC_OK;
    if (!x || !out || n_slabs < 1 || slab_elems < 0) return TAC_E_INVALID;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(ew_blocks(slab_elems)), dim3(EW_THREADS), 0, (hipStream_t)stream, x,
                       (long long)n_slabs, (long long)slab_elems, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // extern "C"
