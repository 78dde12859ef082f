// hpss.hip — harmonic / percussive separation of a magnitude spectrogram by median filtering (reference
// torchaudio_contrib/beta_hpss.py:35-127): the percussive-enhanced spectrogram is the running median
// along frequency, the harmonic-enhanced one the running median along time (reflect padding, `power` applied to both),
// from which soft ((h + eps) / (h + p + eps)) or hard (h > p) masks and the masked spectrograms follow.  The reference
// loops over columns / rows calling torch.median; here one thread owns one (row, f, t) element, gathers the two
// windows (<= 32 taps each, +inf padded), sorts each in registers with Batcher's odd-even merge network (191
// compare-exchanges, fully unrolled) and emits all four outputs in one pass.
#include "host_common.hpp"

namespace tac {

__device__ __forceinline__ void cswap(float& a, float& b) {
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}

// Batcher odd-even mergesort of 32 values held in registers (all indices compile-time after unrolling)
__device__ __forceinline__ void sort32(float (&a)[32]) {
#pragma unroll
    for (int p = 1; p < 32; p *= 2)
#pragma unroll
        for (int k = p; k >= 1; k /= 2)
#pragma unroll
            for (int j = k % p; j + k < 32; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < 32 && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cswap(a[i + j], a[i + j + k]);
}

__device__ __forceinline__ int reflect_index(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__global__ void __launch_bounds__(256)
hpss_kernel(const float* __restrict__ x, long long rows, int F, int T, long long sr, long long sf, long long st, int kf,
            int kt, float power, int hard, float* __restrict__ harm_o, float* __restrict__ perc_o, float* __restrict__ mh_o,
            float* __restrict__ mp_o) {
    const long long per_row = (long long)F * T;
    const long long total = rows * per_row;
    const bool f_fast = sf <= st;                                        // walk the denser axis with consecutive threads
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / per_row;
        const long long rem = idx - row * per_row;
        const int f = f_fast ? (int)(rem % F) : (int)(rem / T);
        const int t = f_fast ? (int)(rem / F) : (int)(rem % T);
        const float* xr = x + row * sr;
        float a[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            a[i] = i < kf ? xr[(long long)reflect_index(f + i - kf / 2, F) * sf + (long long)t * st] : INFINITY;
        sort32(a);
        float perc = a[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) perc = (i == kf / 2) ? a[i] : perc;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            a[i] = i < kt ? xr[(long long)f * sf + (long long)reflect_index(t + i - kt / 2, T) * st] : INFINITY;
        sort32(a);
        float harm = a[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) harm = (i == kt / 2) ? a[i] : harm;
        if (power == 2.0f) {
            perc *= perc;
            harm *= harm;
        } else if (power != 1.0f) {
            perc = powf(perc, power);
            harm = powf(harm, power);
        }
        float mh, mp;
        if (hard) {
            mh = harm > perc ? 1.0f : 0.0f;
            mp = harm < perc ? 1.0f : 0.0f;
        } else {
#pragma clang fp contract(off)
            const float eps = 1e-6f;
            const float den = harm + perc + eps;
            mh = (harm + eps) / den;
            mp = (perc + eps) / den;
        }
        const long long o = row * sr + (long long)f * sf + (long long)t * st;
        const float v = xr[(long long)f * sf + (long long)t * st];
        mh_o[o] = mh;
        mp_o[o] = mp;
        if (harm_o) {
            harm_o[o] = v * mh;
            perc_o[o] = v * mp;
        }
    }
}

}  // namespace tac

extern "C" {

int tac_hpss_f32(const float* mag, int64_t rows, int32_t n_freqs, int32_t n_frames, int64_t stride_r, int64_t stride_f,
                 int64_t stride_t, int32_t kernel_f, int32_t kernel_t, float power, int hard, float* harm, float* perc,
                 float* mask_harm, float* mask_perc, void* stream) {
    using namespace tac;
    if (rows == 0 || n_freqs == 0 || n_frames == 0) return TAC_OK;
    if (!mag || !mask_harm || !mask_perc || (harm == nullptr) != (perc == nullptr)) return TAC_E_INVALID;
    if (rows < 0 || n_freqs < 0 || n_frames < 0) return TAC_E_INVALID;
    if (kernel_f < 1 || kernel_t < 1 || !(kernel_f & 1) || !(kernel_t & 1) || kernel_f > 32 || kernel_t > 32)
        return TAC_E_UNSUPPORTED;
    if (kernel_f / 2 >= n_freqs || kernel_t / 2 >= n_frames) return TAC_E_SHORT_INPUT;      // reflect padding needs pad < size
    const long long total = rows * (long long)n_freqs * n_frames;
    long long blocks = (total + 255) / 256;
    const long long cap = (long long)device_cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(hpss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mag, (long long)rows,
                       (int)n_freqs, (int)n_frames, (long long)stride_r, (long long)stride_f, (long long)stride_t,
                       (int)kernel_f, (int)kernel_t, power, hard, harm, perc, mask_harm, mask_perc);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // extern "C"
