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
#ifndef TAC_PV_HW_SINCOS
#define TAC_PV_HW_SINCOS 1     // 0: library sincosf on the float64-reduced angle (0.293 vs 0.274 ms at cfg-2)
#endif

// Precision.  The reference evaluates the recurrence in the dtype of its input, and in float32 that is
// ill-conditioned: `angle_1 - angle_0 - phase_advance` is rounded at the magnitude of the phase advance (up to
// pi*hop, i.e. several hundred radians: spacing 6e-5) and the running sum at its own magnitude (thousands of
// radians: spacing 5e-4), which is why the reference's own test runs it in float64 (tests/test_functional.py:85-88).
// Here the samples, arctangents, magnitudes and outputs have the precision of T, but the phase increment and the
// running sum are always carried in float64, so a float32 call agrees with the float64 evaluation of the same
// inputs to ~1e-6 instead of ~1e-3; T = double is the reference's float64 path as is.
template <class T>
struct pv_math;
template <>
struct pv_math<float> {
    // atan2f without the library's special-case ladder (~25 instructions instead of ~50; the kernel is bound by this
    // arithmetic, not by memory): octant reduction to a = min / max in [0, 1], atan(a) = a P(a^2) with a degree-8 minimax P
    // (max error 1.0e-7 rad in float32, fitted and checked over 2 M points by the script quoted in tools/ablation/README.md),
    // signs restored.  atan2(0, 0) = 0 like the library; the errors telescope in the running sum (a step's second
    // phase is the next step's first).
    static __device__ __forceinline__ float atan2(float y, float x) {
        const float ax = fabsf(x), ay = fabsf(y);
        const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
        float a = mn * __builtin_amdgcn_rcpf(mx);
        a = mx == 0.0f ? 0.0f : a;
        const float s = a * a;
        float p = 0.0024567253421992064f;
        p = fmaf(p, s, -0.014401361346244812f);
        p = fmaf(p, s, 0.03978123143315315f);
        p = fmaf(p, s, -0.07234857976436615f);
        p = fmaf(p, s, 0.10498946160078049f);
        p = fmaf(p, s, -0.14161229133605957f);
        p = fmaf(p, s, 0.19985906779766083f);
        p = fmaf(p, s, -0.33332598209381104f);
        p = fmaf(p, s, 0.9999998807907104f);
        float r = p * a;
        r = ay > ax ? 1.5707963267948966f - r : r;
        r = x < 0.0f ? 3.141592653589793f - r : r;
        return copysignf(r, y);
    }
    static __device__ __forceinline__ float hypot(float x, float y) { return sqrtf(x * x + y * y); }
    static __device__ __forceinline__ void sincos(double a, float* s, float* c) {
#if TAC_PV_HW_SINCOS
        const double tr = a * 0.15915494309189535;                  // reduced to a fraction of a turn in float64,
        const float fr = (float)(tr - rint(tr));                    // evaluated by the hardware's v_sin / v_cos (input in turns)
        *s = __builtin_amdgcn_sinf(fr);
        *c = __builtin_amdgcn_cosf(fr);
#else
        const double turns = rint(a * 0.15915494309189535);
        sincosf((float)(a - turns * 6.283185307179586), s, c);      // reduced in float64, evaluated in float32
#endif
    }
};
template <>
struct pv_math<double> {
    static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double hypot(double x, double y) { return sqrt(x * x + y * y); }
    static __device__ __forceinline__ void sincos(double a, double* s, double* c) { ::sincos(a, s, c); }
};

template <class T>
__global__ void __launch_bounds__(PV_THREADS)
phase_vocoder_kernel(const T* __restrict__ spec, long long rows, int n_freqs, int n_frames, long long stride_r,
                     long long stride_f, long long stride_t, const T* __restrict__ phase_advance,
                     const int* __restrict__ idx0, const int* __restrict__ idx1, const T* __restrict__ alpha,
                     int n_out, T* __restrict__ out) {
    using M = pv_math<T>;
    typedef T T2 __attribute__((ext_vector_type(2)));      // one (re, im) pair = one 8- / 16-byte access
    // series are numbered row-major over (row, frequency): no partly filled workgroup per row (1025 bins would leave
    // every fifth 256-thread group with a single live lane walking the whole time axis)
    const long long sid = (long long)blockIdx.x * PV_THREADS + threadIdx.x;
    if (sid >= rows * n_freqs) return;
    const long long row = sid / n_freqs;
    const int f = (int)(sid - row * n_freqs);
    const T* base = spec + row * stride_r + (long long)f * stride_f;
    T re0, im0, re1, im1;
    auto frame = [&](int t, T& re, T& im) {     // the two frames past the end are the reference's zero padding
        if (t >= n_frames) {
            re = im = (T)0;
            return;
        }
        const T2 v = *reinterpret_cast<const T2*>(base + (long long)t * stride_t);
        re = v.x;
        im = v.y;
    };
    const double two_pi = 6.283185307179586;
    const double pa = (double)phase_advance[f];
    frame(0, re0, im0);
    double acc = (double)M::atan2(im0, re0);    // phase of the first input frame opens the running sum
    T* o = out + (row * n_out * (long long)n_freqs + f) * 2;
    // The second frame of step i is the first frame of step i + 1 whenever the grid advances by one input frame (every
    // step for rate <= 1, most steps up to rate 2): its phase and magnitude are kept instead of being loaded and
    // evaluated again (the grid is wave-uniform, so is the branch; the values are the ones that would be recomputed).
    int t_kept = -1;
    T ang_kept = (T)0, n_kept = (T)0;
    // The loop is a chain of dependent steps, and a step's only long latency is the load of its second frame: left in
    // the step it made every one of the n_out steps one HBM round trip long (0.336 ms at cfg-2, whatever the occupancy).
    // The second frames of the next PV_AHEAD steps are therefore always in flight (their indices come from the grid,
    // not from the recurrence), in a rotating set of registers.
    constexpr int AHEAD = 4;
    T2 ahead[AHEAD];
    auto fetch = [&](int i) -> T2 {             // second frame of step i (clamped to the last step; zero past the input)
        const int ic = i < n_out ? i : n_out - 1;
        const int t = idx1[ic];
        const int tc = t < n_frames ? t : n_frames - 1;
        T2 v = *reinterpret_cast<const T2*>(base + (long long)tc * stride_t);
        if (t >= n_frames) v.x = v.y = (T)0;
        return v;
    };
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) ahead[k] = fetch(k);
    for (int i0 = 0; i0 < n_out; i0 += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const int i = i0 + k;
            if (i >= n_out) break;
            const int t0 = idx0[i], t1 = idx1[i];
            const T2 cur = ahead[k];
            ahead[k] = fetch(i + AHEAD);
            T ang0, n0;
            if (t0 == t_kept) {
                ang0 = ang_kept;
                n0 = n_kept;
            } else {
                frame(t0, re0, im0);
                ang0 = M::atan2(im0, re0);
                n0 = M::hypot(re0, im0);
            }
            re1 = cur.x;
            im1 = cur.y;
            const T ang1 = M::atan2(im1, re1), n1 = M::hypot(re1, im1);
            t_kept = t1;
            ang_kept = ang1;
            n_kept = n1;
            const T w = alpha[i];
            const T mag = w * n1 + ((T)1 - w) * n0;
            T sn, cs;
            M::sincos(acc, &sn, &cs);
            T2 res;
            res.x = mag * cs;
            res.y = mag * sn;
            *reinterpret_cast<T2*>(o) = res;
            o += 2 * (long long)n_freqs;
            double ph = (double)ang1 - (double)ang0 - pa;
            // (a reciprocal instead of the reference's division: where the two round differently the wrapped phase moves by
            // exactly one turn, which the sine and cosine of the running sum do not see)
            ph = ph - two_pi * rint(ph * 0.15915494309189535);
            acc += ph + pa;
        }
    }
}

template <class T>
static int launch_phase_vocoder(const T* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                                int64_t stride_f, int64_t stride_t, const T* phase_advance, const int32_t* idx0,
                                const int32_t* idx1, const T* alpha, int64_t n_out, T* out, void* stream) {
    if (rows == 0 || n_out == 0 || n_freqs == 0) return TAC_OK;
    if (!spec || !phase_advance || !idx0 || !idx1 || !alpha || !out) return TAC_E_INVALID;
    if (rows < 0 || n_freqs < 0 || n_frames <= 0 || n_out < 0) return TAC_E_INVALID;
    if (n_frames >= 0x7fffffffLL || n_out >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const long long blocks = (rows * (long long)n_freqs + PV_THREADS - 1) / PV_THREADS;
    if (blocks >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    hipLaunchKernelGGL(phase_vocoder_kernel<T>, dim3((unsigned)blocks), dim3(PV_THREADS), 0, (hipStream_t)stream, spec,
                       (long long)rows, (int)n_freqs, (int)n_frames, (long long)stride_r, (long long)stride_f,
                       (long long)stride_t, phase_advance, idx0, idx1, alpha, (int)n_out, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // namespace tac

extern "C" {

int tac_phase_vocoder_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                          int64_t stride_f, int64_t stride_t, const float* phase_advance, const int32_t* idx0,
                          const int32_t* idx1, const float* alpha, int64_t n_out, float* out, void* stream) {
    return tac::launch_phase_vocoder<float>(spec, rows, n_freqs, n_frames, stride_r, stride_f, stride_t, phase_advance,
                                            idx0, idx1, alpha, n_out, out, stream);
}

int tac_phase_vocoder_f64(const double* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                          int64_t stride_f, int64_t stride_t, const double* phase_advance, const int32_t* idx0,
                          const int32_t* idx1, const double* alpha, int64_t n_out, double* out, void* stream) {
    return tac::launch_phase_vocoder<double>(spec, rows, n_freqs, n_frames, stride_r, stride_f, stride_t, phase_advance,
                                             idx0, idx1, alpha, n_out, out, stream);
}

}  // extern "C"
