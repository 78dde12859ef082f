// hpss.hip — harmonic / percussive separation of a magnitude spectrogram by median filtering (reference
// torchaudio_contrib/beta_hpss.py:35-127): the percussive-enhanced spectrogram is the running median
// along frequency, the harmonic-enhanced one the running median along time (reflect padding, `power` applied to both),
// from which soft ((h + eps) / (h + p + eps)) or hard (h > p) masks and the masked spectrograms follow.  The reference
// loops over columns / rows calling torch.median.
//
// hpss_tile_kernel (equal odd widths 9 ... 31 — the reference only runs with equal widths, default 31): a 256-thread
// workgroup owns a 64 x 64 tile of one spectrogram, staged ONCE in LDS with its reflect-padded halo (the fast memory
// axis is the LDS column axis, so the fill is coalesced whatever the layout: contiguous (F, T) or the frame-major
// strided views the STFT kernels return); a thread owns a 4 x 4 block of outputs.  Four consecutive windows along an
// axis share K - 3 of their K taps: those are sorted once in registers (Batcher network) and each window's median is
// selected from five of them and the window's three own taps (median_run.hpp) — 110 min/max operations per median
// instead of the 382 of a full 32-sort per output.  A NaN anywhere in a window makes its median NaN, as torch.median
// does (the min/max network alone would drop it).
//
// hpss_kernel: the general form (unequal or small widths): one thread per element, two full sorts.
#include "host_common.hpp"
#include "median_run.hpp"

namespace tac {

__device__ __forceinline__ int reflect_index(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }
__device__ __forceinline__ int reflect_clamped(int i, int n) {
    const int j = reflect_index(i, n);
    return j < 0 ? 0 : (j >= n ? n - 1 : j);        // (halo positions that only out-of-range outputs would use)
}

constexpr int HP_TILE = 64;                  // outputs per tile side
constexpr int HP_ROWS = HP_TILE + 30;        // slow-axis extent incl. the largest halo (15 each side)
constexpr int HP_LEFT = 16;                  // fast-axis halo on the low side (16: the tile's own columns stay 16-byte aligned)
constexpr int HP_STRIDE = 96;                // 16 + 64 + 15, rounded up to whole 16-byte chunks
constexpr int HP_LDS_FLOATS = HP_ROWS * HP_STRIDE;

typedef float hp_f4 __attribute__((ext_vector_type(4)));
typedef float hp_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void hpss_masks(float harm, float perc, float power, int hard, float& mh, float& mp) {
    if (power == 2.0f) {
        perc *= perc;
        harm *= harm;
    } else if (power != 1.0f) {
        perc = powf(perc, power);
        harm = powf(harm, power);
    }
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
}

// A = the slow memory axis (stride sa), B = the fast one (stride sb); b_is_time says which of them is time.
#ifndef TAC_HPSS_OCC
#define TAC_HPSS_OCC 3   // k = 31: 176 -> 168 registers (36 bytes of scratch) buys a third wave per SIMD: 1.05 -> 0.91 ms; 4 (128 registers, 252 B) 1.26 ms
#endif
template <int K>
__global__ void __launch_bounds__(256, TAC_HPSS_OCC)
hpss_tile_kernel(const float* __restrict__ x, int NA, int NB, long long sr, long long sa, long long sb, int tiles_a,
                 int tiles_b, int b_is_time, float power, int hard, float* __restrict__ harm_o, float* __restrict__ perc_o,
                 float* __restrict__ mh_o, float* __restrict__ mp_o) {
    constexpr int HALF = K / 2;
    __shared__ __attribute__((aligned(16))) float tile[HP_LDS_FLOATS];
    const int tid = threadIdx.x;
    const int per_row = tiles_a * tiles_b;
    const long long row = blockIdx.x / per_row;
    const int rem = (int)(blockIdx.x - row * per_row);
    const int a0 = (rem / tiles_b) * HP_TILE, b0 = (rem % tiles_b) * HP_TILE;
    const float* xr = x + row * sr;
    for (int i = tid; i < HP_LDS_FLOATS; i += 256) {
        const int r = i / HP_STRIDE, c = i - r * HP_STRIDE;
        const int a = reflect_clamped(a0 - 15 + r, NA), b = reflect_clamped(b0 - HP_LEFT + c, NB);
        tile[i] = xr[(long long)a * sa + (long long)b * sb];
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    // a window holds a NaN <=> its median is NaN (torch.median); the min/max network alone would drop it
    auto run4 = [](const float (&w)[K + 3], float (&med)[4]) {
        bool nan_c = false;
#pragma unroll
        for (int u = 3; u < K; ++u) nan_c |= w[u] != w[u];
        median_run4<K>(w, med);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool bad = nan_c;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const float e = w[u < 3 - j ? j + u : K + (u - (3 - j))];
                bad |= e != e;
            }
            med[j] = bad ? __builtin_nanf("") : med[j];
        }
    };
    float medA[4][4];                      // [a][b] of the thread's 4 x 4 block: medians along A
    // ---- along A: two block columns per pass (8-byte LDS reads), one run of four windows per column
#pragma unroll
    for (int jp = 0; jp < 4; jp += 2) {
        float w0[K + 3], w1[K + 3];
        const float* src = tile + (4 * ty + 15 - HALF) * HP_STRIDE + 4 * tx + HP_LEFT + jp;
#pragma unroll
        for (int u = 0; u < K + 3; ++u) {
            const hp_f2 v = *reinterpret_cast<const hp_f2*>(src + u * HP_STRIDE);
            w0[u] = v.x;
            w1[u] = v.y;
        }
        float med[4];
        run4(w0, med);
#pragma unroll
        for (int i = 0; i < 4; ++i) medA[i][jp] = med[i];
        __builtin_amdgcn_sched_barrier(0);
        run4(w1, med);
#pragma unroll
        for (int i = 0; i < 4; ++i) medA[i][jp + 1] = med[i];
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- along B (the fast axis): one run of four windows per block row, then that row's masks and stores
    constexpr int START = HP_LEFT - HALF, OFF = START & 3, NCH = (K + 3 + OFF + 3) / 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const hp_f4* src = reinterpret_cast<const hp_f4*>(tile + (4 * ty + 15 + i) * HP_STRIDE + 4 * tx + (START - OFF));
        float buf[4 * NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const hp_f4 v = src[u];
            buf[4 * u] = v.x; buf[4 * u + 1] = v.y; buf[4 * u + 2] = v.z; buf[4 * u + 3] = v.w;
        }
        float w[K + 3];
#pragma unroll
        for (int u = 0; u < K + 3; ++u) w[u] = buf[OFF + u];
        float medB[4];
        run4(w, medB);
        const int a = a0 + 4 * ty + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + 4 * tx + j;
            if (a < NA && b < NB) {
                const float harm = b_is_time ? medB[j] : medA[i][j];
                const float perc = b_is_time ? medA[i][j] : medB[j];
                float mh, mp;
                hpss_masks(harm, perc, power, hard, mh, mp);
                const long long o = row * sr + (long long)a * sa + (long long)b * sb;
                const float v = w[HALF + j];                           // the window's own centre tap
#ifdef TAC_HPSS_ABL_NOSTORE
                if (mh + mp != 123.0f) continue;
#endif
                mh_o[o] = mh;
                mp_o[o] = mp;
                if (harm_o) {
                    harm_o[o] = v * mh;
                    perc_o[o] = v * mp;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

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
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            a[i] = i < kf ? xr[(long long)reflect_index(f + i - kf / 2, F) * sf + (long long)t * st] : INFINITY;
            bad |= a[i] != a[i];
        }
        sort_net<32>(a);
        float perc = a[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) perc = (i == kf / 2) ? a[i] : perc;
        perc = bad ? __builtin_nanf("") : perc;                          // torch.median: a NaN in the window is the median
        bad = false;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            a[i] = i < kt ? xr[(long long)f * sf + (long long)reflect_index(t + i - kt / 2, T) * st] : INFINITY;
            bad |= a[i] != a[i];
        }
        sort_net<32>(a);
        float harm = a[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) harm = (i == kt / 2) ? a[i] : harm;
        harm = bad ? __builtin_nanf("") : harm;
        float mh, mp;
        hpss_masks(harm, perc, power, hard, mh, mp);
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

template <int K>
static void launch_tile(const float* mag, long long rows, int F, int T, long long sr, long long sf, long long st,
                        float power, int hard, float* harm, float* perc, float* mh, float* mp, hipStream_t stream) {
    const bool t_fast = st <= sf;
    const int NA = t_fast ? F : T, NB = t_fast ? T : F;
    const long long sa = t_fast ? sf : st, sb = t_fast ? st : sf;
    const int ta = (NA + HP_TILE - 1) / HP_TILE, tb = (NB + HP_TILE - 1) / HP_TILE;
    hipLaunchKernelGGL(hpss_tile_kernel<K>, dim3((unsigned)(rows * ta * tb)), dim3(256), 0, stream, mag, NA, NB, sr, sa, sb,
                       ta, tb, t_fast ? 1 : 0, power, hard, harm, perc, mh, mp);
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
    const long long tiles = rows * ((n_freqs + HP_TILE - 1) / HP_TILE) * ((n_frames + HP_TILE - 1) / HP_TILE);
    if (kernel_f == kernel_t && kernel_f >= 9 && tiles < 0x7fffffffLL) {
#define TAC_HPSS_CASE(K)                                                                                                   \
    case K:                                                                                                               \
        launch_tile<K>(mag, rows, n_freqs, n_frames, stride_r, stride_f, stride_t, power, hard, harm, perc, mask_harm,   \
                       mask_perc, (hipStream_t)stream);                                                                  \
        break;
        switch (kernel_f) {
            TAC_HPSS_CASE(9) TAC_HPSS_CASE(11) TAC_HPSS_CASE(13) TAC_HPSS_CASE(15) TAC_HPSS_CASE(17) TAC_HPSS_CASE(19)
            TAC_HPSS_CASE(21) TAC_HPSS_CASE(23) TAC_HPSS_CASE(25) TAC_HPSS_CASE(27) TAC_HPSS_CASE(29) TAC_HPSS_CASE(31)
        }
#undef TAC_HPSS_CASE
        TAC_HIP(hipGetLastError());
        return TAC_OK;
    }
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
