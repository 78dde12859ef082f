// phase_vocoder.hip — time-stretch of a complex spectrogram (functional.py:204-274).
//
// Every (row, frequency) series is an independent recurrence over the output frames — interpolate the two
// neighbouring input magnitudes, advance an accumulated phase by the wrapped phase difference — so one lane owns one
// series and walks the time axis; lanes of a wave are consecutive frequency bins, which makes every load and store a
// contiguous row segment for frame-major spectrograms (the layout the STFT kernels of this library produce; any
// other strides work, just less coalesced).  The fractional time grid (indices of the two source frames and the
// interpolation weight per output frame) is an input: the reference derives it with torch.arange on its own device
// and its float32 rounding decides which frames are paired, so the host wrapper computes it the reference's way.
#include "host_common.hpp"

namespace tac {

constexpr int PV_THREADS = 256;

__global__ void __launch_bounds__(PV_THREADS)
phase_vocoder_kernel(const float* __restrict__ spec, long long rows, int n_freqs, int n_frames, long long stride_r,
                     long long stride_f, long long stride_t, const float* __restrict__ phase_advance,
                     const int* __restrict__ idx0, const int* __restrict__ idx1, const float* __restrict__ alpha,
                     int n_out, float* __restrict__ out) {
    const int fblocks = (n_freqs + PV_THREADS - 1) / PV_THREADS;
    const long long row = blockIdx.x / fblocks;
    const int f = (int)(blockIdx.x % fblocks) * PV_THREADS + threadIdx.x;
    if (row >= rows || f >= n_freqs) return;
    const float* base = spec + row * stride_r + (long long)f * stride_f;
    auto frame = [&](int t) -> cf {            // the two frames past the end are the reference's zero padding
        if (t >= n_frames) return mkc(0.0f, 0.0f);
        const float* p = base + (long long)t * stride_t;
        return mkc(p[0], p[1]);
    };
    const float two_pi = 6.283185307179586f;   // float32(2*math.pi), as the reference's scalar ops see it
    const float pa = phase_advance[f];
    const cf z0 = frame(0);
    float acc = atan2f(z0.y, z0.x);            // phase of the first input frame opens the running sum
    float* o = out + (row * n_out * (long long)n_freqs + f) * 2;
    for (int i = 0; i < n_out; ++i) {
#pragma clang fp contract(off)                 // the reference rounds every product before it adds: keep its op sequence
        const cf a = frame(idx0[i]), b = frame(idx1[i]);
        const float n0 = sqrtf(a.x * a.x + a.y * a.y), n1 = sqrtf(b.x * b.x + b.y * b.y);
        const float w = alpha[i];
        const float mag = w * n1 + (1.0f - w) * n0;
        float sn, cs;
        sincosf(acc, &sn, &cs);
        *reinterpret_cast<cf*>(o) = mkc(mag * cs, mag * sn);
        o += 2 * (long long)n_freqs;
        float ph = atan2f(b.y, b.x) - atan2f(a.y, a.x) - pa;
        ph = ph - two_pi * rintf(ph / two_pi);
        acc += ph + pa;
    }
}

}  // namespace tac

extern "C" {

int tac_phase_vocoder_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                          int64_t stride_f, int64_t stride_t, const float* phase_advance, const int32_t* idx0,
                          const int32_t* idx1, const float* alpha, int64_t n_out, float* out, void* stream) {
    using namespace tac;
    if (rows == 0 || n_out == 0 || n_freqs == 0) return TAC_OK;
    if (!spec || !phase_advance || !idx0 || !idx1 || !alpha || !out) return TAC_E_INVALID;
    if (rows < 0 || n_freqs < 0 || n_frames <= 0 || n_out < 0) return TAC_E_INVALID;
    if (n_frames >= 0x7fffffffLL || n_out >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const long long fblocks = (n_freqs + PV_THREADS - 1) / PV_THREADS;
    const long long blocks = rows * fblocks;
    if (blocks >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    hipLaunchKernelGGL(phase_vocoder_kernel, dim3((unsigned)blocks), dim3(PV_THREADS), 0, (hipStream_t)stream, spec,
                       (long long)rows, (int)n_freqs, (int)n_frames, (long long)stride_r, (long long)stride_f,
                       (long long)stride_t, phase_advance, idx0, idx1, alpha, (int)n_out, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // extern "C"
