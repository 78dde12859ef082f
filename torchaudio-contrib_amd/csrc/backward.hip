// backward.hip — gradient kernels of the path.  The reference differentiates through stock
// torch ops (functional.py:99-107 stft, :126-128 norm/pow, :183-184 matmul, :291-296 dB); these are the adjoints
// of this library's forward kernels, written against the same frame-major layouts:
//
//   tac_stft_backward_f32        g_spec[rows][T][F][2] -> windowed frame gradients [rows][T][N]: per frame one inverse
//                                real FFT of the one-sided gradient spectrum (the C2R form of the forward R2C split,
//                                evaluated with the SAME wave-level forward FFT on conjugated data), times the window.
//   tac_overlap_add_f32          frame gradients -> g_wave[rows][L]: the adjoint of framing + padding as a GATHER
//                                (every output sample sums the <= 3 padded positions that map to it over the frames
//                                covering them), so it is deterministic and needs no atomics.
//   tac_complex_norm_backward_f32, tac_amplitude_to_db_backward_f32   elementwise.
// The filterbank stage's adjoint is the forward GEMM with the transposed matrix (tac_apply_filterbank_f32).
#include "host_common.hpp"

namespace tac {

constexpr int BW_WAVES = 4;

// One frame (group) per wave-iteration.  y[n] = Re sum_{k=0}^{N/2} G[k] e^{+2 pi i k n / N}:
//   H[k] = G[k] (0 < k < NC), H[0] = 2 Re G[0], H[NC] = 2 Re G[NC]            (the common 1/2 is folded into the window)
//   conj(Z[k]) = (conj(H[k]) + H[NC-k]) - i w_k (conj(H[k]) - H[NC-k]),  w_k = e^{-2 pi i k / N}
//   R = FFT_NC(conj Z);  y[2m] = Re R[m],  y[2m+1] = -Im R[m].
// d/dz of |z|^power (norm then pow, functional.py:126-128): g * power * |z|^(power-2) * z, 0 at z == 0
__device__ __forceinline__ cf norm_pow_grad(cf v, float gout, float power) {
    const float s = v.x * v.x + v.y * v.y;
    float f;
    if (power == 2.0f) f = 2.0f;
    else if (s == 0.0f) f = 0.0f;
    else if (power == 1.0f) f = 1.0f / sqrtf(s);
    else f = power * powf(sqrtf(s), power - 2.0f);
    f *= gout;
    return mkc(f * v.x, f * v.y);
}

// NORM: `gspec` is the spectrum z itself and `gnorm` the gradient of |z|^power: the gradient spectrum
// gnorm * d|z|^power/dz is formed on load (tac_stft_norm_backward_f32: the adjoint of Spectrogram in one pass, no
// gradient spectrum in memory)
template <int NC, int E, bool NORM>
__global__ void __launch_bounds__(BW_WAVES * 64, 2)
stft_backward_kernel(FrameGeom g, Tables tb, const float* __restrict__ gspec, const float* __restrict__ gnorm, float power,
                     float* __restrict__ frames) {
    using F = WaveFft<NC, E>;
    constexpr int N = 2 * NC, NBINS = NC + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / F::LPF;
    const int t = lane % F::LPF;
    constexpr int WAVE_SLOTS = ((F::G * F::PADDED + 1) / 2) * 2;
    cf* lds = reinterpret_cast<cf*>(smem_raw) + w * WAVE_SLOTS + sub * F::PADDED;

    const long long groups_per_row = (g.n_frames + F::G - 1) / F::G;
    const long long total = g.rows * groups_per_row;
    const float wscale = 0.5f * g.scale;
    constexpr int R0 = radix_at(NC, 0), NB = E / R0;
    // lane-dependent tables stay in registers for the kernel's lifetime where they fit (16 elements per lane): the FFT's
    // inter-pass twiddles, the C2R twiddles of the lane's sixteen (k, NC - k) pairs, its sixteen window pairs
    constexpr bool HOIST = (E == 16);
    constexpr bool HOIST_WIN = HOIST && !NORM;             // (the NORM form keeps 32 more loads in flight per frame)
    cf tw_h[HOIST ? F::NTW : 1], wk_h[HOIST ? E : 1], win_h[HOIST_WIN ? E : 1];
    if constexpr (HOIST) {
        F::load_twiddles(tw_h, tb.w_nc, t);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const int k = t + b * F::LPF + q * (NC / R0);
                const cf wk = tb.w_n[k <= NC / 2 ? k : NC - k];            // w_{NC-k} = -conj(w_k)
                wk_h[b * R0 + q] = k <= NC / 2 ? wk : mkc(-wk.x, wk.y);
            }
        if constexpr (HOIST_WIN) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const cf wn = window_pair(g, t + j * F::LPF);
                win_h[j] = mkc(wn.x * wscale, -wn.y * wscale);
            }
        }
    }
    for (long long unit = (long long)blockIdx.x * BW_WAVES + w; unit < total; unit += (long long)gridDim.x * BW_WAVES) {
        const long long row = unit / groups_per_row;
        const long long frame = (unit - row * groups_per_row) * F::G + sub;
        const bool live = frame < g.n_frames;
        const cf* G = reinterpret_cast<const cf*>(gspec) + (row * g.n_frames + (live ? frame : 0)) * NBINS;
        const float* GN = NORM ? gnorm + (row * g.n_frames + (live ? frame : 0)) * NBINS : nullptr;
        cf tw_l[HOIST ? 1 : F::NTW];
        if constexpr (!HOIST) F::load_twiddles(tw_l, tb.w_nc, t);
        const cf* const tw = HOIST ? tw_h : tw_l;
        cf v[1][E];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const int k = t + b * F::LPF + q * (NC / R0);              // first-pass order (fft_core.hpp)
                cf hk = G[k], hm = G[NC - k];
                if constexpr (NORM) {
                    hk = norm_pow_grad(hk, GN[k], power);
                    hm = norm_pow_grad(hm, GN[NC - k], power);
                }
                if (k == 0) {
                    hk = mkc(2.0f * hk.x, 0.0f);
                    hm = mkc(2.0f * hm.x, 0.0f);
                }
                if (!live) hk = hm = mkc(0.0f, 0.0f);
                cf wkk;
                if constexpr (HOIST) {
                    wkk = wk_h[b * R0 + q];
                } else {
                    const cf wk = tb.w_n[k <= NC / 2 ? k : NC - k];        // w_{NC-k} = -conj(w_k)
                    wkk = k <= NC / 2 ? wk : mkc(-wk.x, wk.y);
                }
                const cf s = mkc(hk.x + hm.x, hm.y - hk.y);                 // conj(H[k]) + H[NC-k]
                const cf d = mkc(hk.x - hm.x, -hk.y - hm.y);                // conj(H[k]) - H[NC-k]
                const cf wd = mkc(wkk.x * d.x - wkk.y * d.y, wkk.x * d.y + wkk.y * d.x);
                v[0][b * R0 + q] = mkc(s.x + wd.y, s.y - wd.x);             // s - i * (w d)
            }
        cf* const ldsv[1] = {lds};
        F::template run<1>(v, ldsv, tw, t);                                 // R[] in natural order at lds[lds_pad(i)]
        float* out = frames + (row * g.n_frames + frame) * N;
        if (live) {
            if constexpr (HOIST_WIN) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const int m = t + j * F::LPF;
                    const cf r = lds[lds_pad(m)];
                    *reinterpret_cast<cf*>(out + 2 * m) = mkc(r.x * win_h[j].x, r.y * win_h[j].y);
                }
            } else {
#pragma unroll 4
                for (int m = t; m < NC; m += F::LPF) {
                    const cf r = lds[lds_pad(m)];
                    const cf wn = window_pair(g, m);
                    *reinterpret_cast<cf*>(out + 2 * m) = mkc(r.x * wn.x * wscale, -r.y * wn.y * wscale);
                }
            }
        }
        wave_lds_fence();
    }
}

// g_wave[row][j] = sum over padded positions i with source(i) == j of sum over frames t covering i of
// frames[row][t][i + pad - t*hop]   (source(): torch.nn.functional.pad semantics, fft_core.hpp padded_index).
__global__ void __launch_bounds__(256)
overlap_add_kernel(FrameGeom g, int n_fft, const float* __restrict__ frames, float* __restrict__ gwave,
                   long long gwave_row_stride) {
    // positions inside a row are 32-bit (L < 2^31 - 2 n_fft, host-checked): only the final addresses are 64-bit — the
    // 64-bit divisions of the first version were most of this kernel's time
    const int L = (int)g.length, T = (int)g.n_frames;
    const int pad = g.center_pad, hop = g.hop;
    const long long total = g.rows * (long long)L;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / L);
        const int j = (int)(idx - (long long)row * L);
        const float* fr = frames + (long long)row * T * n_fft;
        float acc = 0.0f;
        auto add_position = [&](int i) {                                // i: position in the padded signal minus pad
            const int p = i + pad;                                      // 0 <= p < L + 2 pad
            int t1 = p / hop;
            if (t1 > T - 1) t1 = T - 1;
            int t0 = p - n_fft + 1 <= 0 ? 0 : (p - n_fft + hop) / hop;  // ceil((p - n_fft + 1) / hop)
            for (int tt = t0; tt <= t1; ++tt) acc += fr[(long long)tt * n_fft + (p - tt * hop)];
        };
        add_position(j);
        if (pad > 0) {
            if (g.pad_mode == PAD_REFLECT) {
                if (j >= 1 && j <= pad) add_position(-j);
                if (j <= L - 2 && j >= L - 1 - pad) add_position(2 * (L - 1) - j);
            } else if (g.pad_mode == PAD_REPLICATE) {
                if (j == 0) for (int i = -pad; i < 0; ++i) add_position(i);
                if (j == L - 1) for (int i = L; i < L + pad; ++i) add_position(i);
            } else if (g.pad_mode == PAD_CIRCULAR) {
                if (j >= L - pad) add_position(j - L);
                if (j < pad) add_position(j + L);
            }
        }
        gwave[row * gwave_row_stride + j] = acc;
    }
}

// d/dz of |z|^power lives above (norm_pow_grad)
__global__ void __launch_bounds__(256)
complex_norm_backward_kernel(const float* __restrict__ z, const float* __restrict__ gout, long long n, float power,
                             float* __restrict__ gz) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        *reinterpret_cast<cf*>(gz + 2 * i) = norm_pow_grad(*reinterpret_cast<const cf*>(z + 2 * i), gout[i], power);
}

// d/dx of 10 (log10(clamp(x^2, amin)) - log10 ref) (functional.py:291-296): 20 / (ln 10 * x) where x^2 >= amin, else 0
__global__ void __launch_bounds__(256)
amplitude_to_db_backward_kernel(const float* __restrict__ x, const float* __restrict__ gout, long long n, float amin,
                                float* __restrict__ gx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        gx[i] = (v * v >= amin) ? gout[i] * (8.6858896380650366f / v) : 0.0f;
    }
}

template <int NC, int E>
static int launch_stft_backward(const FrameGeom& g, const Tables& tb, const float* gspec, const float* gnorm, float power,
                                float* frames, hipStream_t stream) {
    using F = WaveFft<NC, E>;
    const size_t lds_bytes = (size_t)BW_WAVES * (((F::G * F::PADDED + 1) / 2) * 2) * sizeof(cf);
    const long long groups = g.rows * ((g.n_frames + F::G - 1) / F::G);
    long long blocks = (groups + BW_WAVES - 1) / BW_WAVES;
    const long long cap = (long long)device_cu_count() * 2;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    auto kern = gnorm ? stft_backward_kernel<NC, E, true> : stft_backward_kernel<NC, E, false>;
    if (lds_bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BW_WAVES * 64), lds_bytes, stream, g, tb, gspec, gnorm, power, frames);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

static int stft_backward_entry(const float* spec, const float* gnorm, float power, const float* window, const tac_stft_desc* d,
                               float* grad_frames, void* stream) {
    if (!spec || !grad_frames || !d) return TAC_E_INVALID;
    if (!d->onesided) return TAC_E_UNSUPPORTED;
    FrameGeom g;
    int64_t T = 0;
    // the geometry helper wants a waveform pointer for its alignment flags only; the spectrum stands in
    int rc = make_geometry(spec, window, d, &g, &T);
    if (rc != TAC_OK) return rc;
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (d->n_fft) {
        case 32: return launch_stft_backward<16, 16>(g, tb, spec, gnorm, power, grad_frames, s);
        case 64: return launch_stft_backward<32, 16>(g, tb, spec, gnorm, power, grad_frames, s);
        case 128: return launch_stft_backward<64, 16>(g, tb, spec, gnorm, power, grad_frames, s);
        case 256: return launch_stft_backward<128, 16>(g, tb, spec, gnorm, power, grad_frames, s);
        case 512: return launch_stft_backward<256, 16>(g, tb, spec, gnorm, power, grad_frames, s);
        case 1024: return launch_stft_backward<512, 16>(g, tb, spec, gnorm, power, grad_frames, s);
        case 2048: return launch_stft_backward<1024, 16>(g, tb, spec, gnorm, power, grad_frames, s);
        case 4096: return launch_stft_backward<2048, 32>(g, tb, spec, gnorm, power, grad_frames, s);
        default: return TAC_E_UNSUPPORTED;
    }
}

static unsigned bw_blocks(long long n) {
    long long want = (n + 255) / 256, cap = (long long)device_cu_count() * 16;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

}  // namespace tac

extern "C" {

int tac_stft_backward_f32(const float* grad_spec, const float* window, const tac_stft_desc* d, float* grad_frames,
                          void* stream) {
    return tac::stft_backward_entry(grad_spec, nullptr, 0.0f, window, d, grad_frames, stream);
}

int tac_stft_norm_backward_f32(const float* spec, const float* grad_norm, float power, const float* window,
                               const tac_stft_desc* d, float* grad_frames, void* stream) {
    if (!grad_norm) return TAC_E_INVALID;
    return tac::stft_backward_entry(spec, grad_norm, power, window, d, grad_frames, stream);
}

int tac_overlap_add_f32(const float* grad_frames, const tac_stft_desc* d, float* grad_wave, int64_t grad_row_stride,
                        void* stream) {
    using namespace tac;
    if (!grad_frames || !grad_wave || !d) return TAC_E_INVALID;
    FrameGeom g;
    int64_t T = 0;
    float dummy_window = 0.0f;
    int rc = make_geometry(grad_frames, &dummy_window, d, &g, &T);
    if (rc != TAC_OK) return rc;
    hipLaunchKernelGGL(overlap_add_kernel, dim3(bw_blocks(g.rows * g.length)), dim3(256), 0, (hipStream_t)stream, g,
                       (int)d->n_fft, grad_frames, grad_wave, (long long)grad_row_stride);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_complex_norm_backward_f32(const float* z, const float* grad_out, int64_t n, float power, float* grad_z,
                                  void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!z || !grad_out || !grad_z || n < 0) return TAC_E_INVALID;
    hipLaunchKernelGGL(complex_norm_backward_kernel, dim3(bw_blocks(n)), dim3(256), 0, (hipStream_t)stream, z, grad_out,
                       (long long)n, power, grad_z);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_amplitude_to_db_backward_f32(const float* x, const float* grad_out, int64_t n, float amin, float* grad_x,
                                     void* stream) {
    using namespace tac;
    if (n == 0) return TAC_OK;
    if (!x || !grad_out || !grad_x || n < 0) return TAC_E_INVALID;
    hipLaunchKernelGGL(amplitude_to_db_backward_kernel, dim3(bw_blocks(n)), dim3(256), 0, (hipStream_t)stream, x,
                       grad_out, (long long)n, amin, grad_x);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // extern "C"
