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
#ifndef TAC_PV_AHEAD
#define TAC_PV_AHEAD 4         // steps whose frames are in flight (8: no gain; 12 / 16: the register set spills, +60 %)
#endif

// Precision.  The reference evaluates the recurrence in the dtype of its input, and in float32 that is
// ill-conditioned: `angle_1 - angle_0 - phase_advance` is rounded at the magnitude of the phase advance (up to
// pi*hop, i.e. several hundred radians: spacing 6e-5) and the running sum at its own magnitude (thousands of
// radians: spacing 5e-4), which is why the reference's own test runs it in float64 (tests/test_functional.py:85-88).
// Only the running sum MODULO one turn reaches the output (it goes through cos / sin, functional.py:268-272), and modulo
// one turn the reference's step `wrap(a1 - a0 - pa) + pa` (functional.py:258-264) is `a1 - a0`: the wrap subtracts whole
// turns and the phase advance cancels.  The float32 kernel therefore never forms the ill-conditioned sum: it carries
// exp(i phase) as a unit phasor (the 32-bit-fraction-of-a-turn form of round 4 is in tools/ablation/lab_knobs_r06.patch); it agrees with the float64 evaluation of the reference's formula to ~1e-6 rad, whatever
// the number of steps and the size of the phase advance.  A phase advance that is not finite poisons every step after the
// first, as the reference's cumulative sum does.  T = double is the reference's float64 formula as is.
template <class T>
struct pv_math;
template <>
struct pv_math<float> {
    // exp(i acc) itself is carried: exp(i (acc + a1 - a0)) = exp(i acc) * u1 * conj(u0) with u = z / |z| — four products per
    // factor, one v_rsq_f32 per input frame, and a first-order renormalisation of the running phasor (|u|^2 is within 1e-6 of
    // 1: u *= 1.5 - 0.5 |u|^2) instead of atan2 (25 instructions), v_sin and v_cos.  Rounding: ~1e-7 rad per step, unbiased
    // (1.5e-6 rad after 241 steps; the arctangent form: 1e-7 per angle).  z = 0 has phase 0 like atan2(0, 0); components are
    // rescaled by 2^+-90 before squaring when their larger one is outside [2^-60, 2^60], so |z|^2 neither underflows nor overflows.
    typedef float ang_t __attribute__((ext_vector_type(2)));        // unit phasor
    typedef ang_t acc_t;
    static __device__ __forceinline__ void polar(float re, float im, ang_t& u, float& n) {
        const float m = fmaxf(fabsf(re), fabsf(im));
        const bool tiny = m < 8.673617379884035e-19f, huge = m > 1.152921504606847e18f;          // 2^-60, 2^60
        const float up = tiny ? 1.2379400392853803e27f : (huge ? 8.077935669463161e-28f : 1.0f);  // 2^90, 2^-90
        const float dn = tiny ? 8.077935669463161e-28f : (huge ? 1.2379400392853803e27f : 1.0f);
        // an infinite component: the angle atan2 gives it (0, +-pi/2, pi, +-pi/4, +-3 pi/4: infinite components count as +-1,
        // finite ones as 0) and an infinite magnitude — inf * 2^-90 = inf, rsq(inf) = 0 and inf * 0 would make the PHASOR NaN and
        // poison every later frame of the bin, where the reference only loses the frames that touch the value
        const bool inf = m == __builtin_inff();
        const float x = inf ? (fabsf(re) == m ? copysignf(1.0f, re) : 0.0f) : re * up;
        const float y = inf ? (fabsf(im) == m ? copysignf(1.0f, im) : 0.0f) : im * up;
        const float n2 = fmaf(x, x, y * y);
        const float r = __builtin_amdgcn_rsqf(n2);
        const bool zero = !(n2 > 0.0f) && n2 == n2;                // (+0: the reference's atan2(0, 0) = 0; NaN stays NaN)
        u.x = zero ? 1.0f : x * r;
        u.y = zero ? 0.0f : y * r;
        n = zero ? 0.0f : (inf ? m : (n2 * r) * dn);
    }
    static __device__ __forceinline__ acc_t open(ang_t a) { return a; }
    static __device__ __forceinline__ acc_t step(acc_t acc, ang_t a1, ang_t a0, float) {
        ang_t w, v;
        w.x = fmaf(a1.x, a0.x, a1.y * a0.y);                        // a1 * conj(a0)
        w.y = fmaf(a1.y, a0.x, -(a1.x * a0.y));
        v.x = fmaf(acc.x, w.x, -(acc.y * w.y));
        v.y = fmaf(acc.x, w.y, acc.y * w.x);
        const float g = fmaf(-0.5f, fmaf(v.x, v.x, v.y * v.y), 1.5f);
        v.x *= g;
        v.y *= g;
        return v;
    }
    static __device__ __forceinline__ float poison(float pa) { return pa - pa; }     // 0, or NaN for a non-finite advance
    static __device__ __forceinline__ void sincos(acc_t acc, float bias, float* s, float* c) {
        *c = acc.x + bias;
        *s = acc.y + bias;
    }
};
template <>
struct pv_math<double> {
    typedef double ang_t;
    typedef double acc_t;
    static __device__ __forceinline__ void polar(double re, double im, ang_t& a, double& n) {
        a = ::atan2(im, re);
        n = sqrt(re * re + im * im);
    }
    static __device__ __forceinline__ acc_t open(ang_t a) { return a; }
    static __device__ __forceinline__ acc_t step(acc_t acc, ang_t a1, ang_t a0, double pa) {
        double ph = a1 - a0 - pa;
        // (a reciprocal instead of the reference's division: where the two round differently the wrapped phase moves by
        // exactly one turn, which the sine and cosine of the running sum do not see)
        ph = ph - 6.283185307179586 * rint(ph * 0.15915494309189535);
        return acc + (ph + pa);
    }
    static __device__ __forceinline__ double poison(double) { return 0.0; }
    static __device__ __forceinline__ void sincos(acc_t a, double, double* s, double* c) { ::sincos(a, s, c); }
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
    auto frame = [&](int t) -> T2 {             // the two frames past the end are the reference's zero padding
        const int tc = t < n_frames ? t : n_frames - 1;
        T2 v = *reinterpret_cast<const T2*>(base + (long long)tc * stride_t);
        if (t >= n_frames) v.x = v.y = (T)0;
        return v;
    };
    const T pa = phase_advance[f];
    const T2 first = frame(0);
    typename M::ang_t ang_first;
    T n_first;
    M::polar(first.x, first.y, ang_first, n_first);
    typename M::acc_t acc = M::open(ang_first);             // phase of the first input frame opens the running sum
    T bias = (T)0;
    T* o = out + (row * n_out * (long long)n_freqs + f) * 2;
    // The second frame of step i is the first frame of step i + 1 whenever the grid advances by one input frame (every
    // step for rate <= 1, most steps up to rate 2): its phase and magnitude are kept instead of being loaded and
    // evaluated again (the grid is wave-uniform, so is the branch; the values are the ones that would be recomputed).
    int t_kept = -1;
    typename M::ang_t ang_kept = ang_first;
    T n_kept = (T)0;
    // The loop is a chain of dependent steps, and a step's only long latency is the load of its frames: left in the
    // step it made every one of the n_out steps one HBM round trip long (0.336 ms at cfg-2, whatever the occupancy).
    // The frames of the next AHEAD steps are therefore always in flight (their indices come from the grid, not from the
    // recurrence), in a rotating set of registers: the second frame always; the first one, when it is not the previous
    // step's second (rate > 1), is loaded inside the step.
    constexpr int AHEAD = sizeof(T) == 8 ? 4 : TAC_PV_AHEAD;        // (float64: 16-byte pairs, round 3's depth)
    T2 ahead1[AHEAD], ahead0[AHEAD];
    auto request = [&](int i, T2& v0, T2& v1) {  // frames of step i (clamped to the last step)
        const int ic = i < n_out ? i : n_out - 1;
        v1 = frame(idx1[ic]);
    };
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) {
        ahead0[k] = first;
        request(k, ahead0[k], ahead1[k]);
    }
    for (int i0 = 0; i0 < n_out; i0 += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const int i = i0 + k;
            if (i >= n_out) break;
            const int t0 = idx0[i], t1 = idx1[i];
            const T2 cur1 = ahead1[k];
            T2 cur0 = ahead0[k];
            request(i + AHEAD, ahead0[k], ahead1[k]);
            typename M::ang_t ang0;
            T n0;
            if (t0 == t_kept) {
                ang0 = ang_kept;
                n0 = n_kept;
            } else {
                cur0 = frame(t0);
                M::polar(cur0.x, cur0.y, ang0, n0);
            }
            typename M::ang_t ang1;
            T n1;
            M::polar(cur1.x, cur1.y, ang1, n1);
            t_kept = t1;
            ang_kept = ang1;
            n_kept = n1;
            const T w = alpha[i];
            const T mag = w * n1 + ((T)1 - w) * n0;
            T sn, cs;
            M::sincos(acc, bias, &sn, &cs);
            T2 res;
            res.x = mag * cs;
            res.y = mag * sn;
            *reinterpret_cast<T2*>(o) = res;
            o += 2 * (long long)n_freqs;
            acc = M::step(acc, ang1, ang0, pa);
            if (i == 0) bias = M::poison(pa);
        }
    }
}

// Gradient of the float32 recurrence with respect to the spectrogram (round 6).  out_j = mag_j e^{i acc_j}, mag_j = alpha_j |s1_j| +
// (1 - alpha_j) |s0_j|, acc_j = angle(first) + sum_{i<j} (angle(s1_i) - angle(s0_i)) modulo whole turns (the wrap and the phase advance
// have no derivative: d/d(phase_advance) = 0, as torch.autograd finds through torch.round).  With g_j the incoming gradient:
//   g_mag_j = Re(conj(u_j) g_j),  g_acc_j = mag_j Im(conj(u_j) g_j),  u_j = e^{i acc_j};
//   d/d angle(s1_i) = G_i = sum_{j>i} g_acc_j,  d/d angle(s0_i) = -G_i,  d/d angle(first) = sum_j g_acc_j;
//   d/d|s1_j| = alpha_j g_mag_j,  d/d|s0_j| = (1 - alpha_j) g_mag_j;   d|z|/dz = z / |z|,  d angle(z)/dz = (-im, re) / |z|^2 (0 at z = 0).
// One lane owns one (row, frequency) series, like the forward kernel, and walks it twice with the forward recurrence: the first walk sums
// g_acc, the second forms the suffix sums as total - prefix and adds every step's two contributions to the (zero-initialised, frame-major)
// gradient — plain read-modify-writes: no other lane touches the series.  The zero padding past the last frame takes no gradient.
__global__ void __launch_bounds__(PV_THREADS)
phase_vocoder_backward_kernel(const float* __restrict__ spec, long long rows, int n_freqs, int n_frames, long long stride_r,
                              long long stride_f, long long stride_t, const int* __restrict__ idx0, const int* __restrict__ idx1,
                              const float* __restrict__ alpha, int n_out, const float* __restrict__ gout, float* __restrict__ gspec) {
    using M = pv_math<float>;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const long long sid = (long long)blockIdx.x * PV_THREADS + threadIdx.x;
    if (sid >= rows * n_freqs) return;
    const long long row = sid / n_freqs;
    const int f = (int)(sid - row * n_freqs);
    const float* base = spec + row * stride_r + (long long)f * stride_f;
    auto frame = [&](int t) -> f2 {
        const int tc = t < n_frames ? t : n_frames - 1;
        f2 v = *reinterpret_cast<const f2*>(base + (long long)tc * stride_t);
        if (t >= n_frames) v.x = v.y = 0.0f;
        return v;
    };
    const float* g = gout + (row * n_out * (long long)n_freqs + f) * 2;
    float* gs = gspec + (row * n_frames * (long long)n_freqs + f) * 2;
    const long long step_elems = 2 * (long long)n_freqs;
    const f2 first = frame(0);
    M::ang_t u_first;
    float n_first;
    M::polar(first.x, first.y, u_first, n_first);
    // walk 1: the sum of g_acc over the series
    float total = 0.0f;
    {
        M::acc_t acc = M::open(u_first);
        for (int i = 0; i < n_out; ++i) {
            const f2 z0 = frame(idx0[i]), z1 = frame(idx1[i]);
            M::ang_t u0, u1;
            float n0, n1;
            M::polar(z0.x, z0.y, u0, n0);
            M::polar(z1.x, z1.y, u1, n1);
            const float w = alpha[i], mag = w * n1 + (1.0f - w) * n0;
            const f2 gi = *reinterpret_cast<const f2*>(g + i * step_elems);
            total += mag * (acc.x * gi.y - acc.y * gi.x);
            acc = M::step(acc, u1, u0, 0.0f);
        }
    }
    auto add = [&](int t, M::ang_t u, float n, float g_n, float g_th) {      // gradient through |z| and angle(z) of source frame t
        if (t >= n_frames) return;
        const float r = n > 0.0f ? g_th / n : 0.0f;
        f2* p = reinterpret_cast<f2*>(gs + (long long)t * step_elems);
        f2 v = *p;
        v.x += (n > 0.0f ? g_n * u.x : 0.0f) - r * u.y;
        v.y += (n > 0.0f ? g_n * u.y : 0.0f) + r * u.x;
        *p = v;
    };
    // walk 2: suffix sums as total - prefix, contributions of every step's two source frames
    {
        M::acc_t acc = M::open(u_first);
        float prefix = 0.0f;
        for (int i = 0; i < n_out; ++i) {
            const int t0 = idx0[i], t1 = idx1[i];
            const f2 z0 = frame(t0), z1 = frame(t1);
            M::ang_t u0, u1;
            float n0, n1;
            M::polar(z0.x, z0.y, u0, n0);
            M::polar(z1.x, z1.y, u1, n1);
            const float w = alpha[i], mag = w * n1 + (1.0f - w) * n0;
            const f2 gi = *reinterpret_cast<const f2*>(g + i * step_elems);
            const float g_mag = acc.x * gi.x + acc.y * gi.y;
            prefix += mag * (acc.x * gi.y - acc.y * gi.x);
            const float suffix = total - prefix;                            // sum over the steps after this one
            add(t0, u0, n0, (1.0f - w) * g_mag, -suffix);
            add(t1, u1, n1, w * g_mag, suffix);
            acc = M::step(acc, u1, u0, 0.0f);
        }
    }
    add(0, u_first, n_first, 0.0f, total);                                 // the phase of the first frame opens every running sum
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

int tac_phase_vocoder_backward_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                                   int64_t stride_f, int64_t stride_t, const int32_t* idx0, const int32_t* idx1, const float* alpha,
                                   int64_t n_out, const float* grad_out, float* grad_spec, void* stream) {
    using namespace tac;
    if (rows == 0 || n_out == 0 || n_freqs == 0) return TAC_OK;
    if (!spec || !idx0 || !idx1 || !alpha || !grad_out || !grad_spec) return TAC_E_INVALID;
    if (rows < 0 || n_freqs < 0 || n_frames <= 0 || n_out < 0) return TAC_E_INVALID;
    if (n_frames >= 0x7fffffffLL || n_out >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const long long blocks = (rows * (long long)n_freqs + PV_THREADS - 1) / PV_THREADS;
    if (blocks >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    hipLaunchKernelGGL(phase_vocoder_backward_kernel, dim3((unsigned)blocks), dim3(PV_THREADS), 0, (hipStream_t)stream, spec,
                       (long long)rows, (int)n_freqs, (int)n_frames, (long long)stride_r, (long long)stride_f, (long long)stride_t,
                       idx0, idx1, alpha, (int)n_out, grad_out, grad_spec);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_phase_vocoder_f64(const double* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                          int64_t stride_f, int64_t stride_t, const double* phase_advance, const int32_t* idx0,
                          const int32_t* idx1, const double* alpha, int64_t n_out, double* out, void* stream) {
    return tac::launch_phase_vocoder<double>(spec, rows, n_freqs, n_frames, stride_r, stride_f, stride_t, phase_advance,
                                             idx0, idx1, alpha, n_out, out, stream);
}

}  // extern "C"
