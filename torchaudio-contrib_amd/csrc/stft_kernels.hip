// stft_kernels.hip — framed, windowed, batched R2C FFT with complex / |.|^p (/dB) epilogues.
//
// Replaces torch.stft (+ torch.norm/pow, + amplitude_to_db) on the reference path
// (torchaudio_contrib/functional.py:99-107, 126-128, 291-296).  One wave = G frames; each
// 4-wave workgroup walks a contiguous chunk of (row, frame) groups so the 4x overlap between
// consecutive frames (hop = N/4) is served by L1/L2 and every input sample leaves HBM once.
// Reflect / constant / replicate / circular padding is done by index arithmetic on the unpadded
// waveform — no padded copy, no framed copy, no separate window multiply.
#include "host_common.hpp"

namespace tac {

constexpr int STFT_WAVES = 4;

struct StftEpilogue {
    float* out;
    int onesided;
    int mode;        // 0 complex, 1 magnitude^power
    float power;
    int db;
    float amin;
    float log10_ref;
};

template <int NC, int MODE>
__device__ __forceinline__ void emit_bin(const StftEpilogue& ep, float* obase, int bin, cf x, float scale) {
    constexpr int N = 2 * NC;
    x.x *= scale;
    x.y *= scale;
    if constexpr (MODE == 0) {
        reinterpret_cast<float2*>(obase)[bin] = x;
        if (!ep.onesided && bin > 0 && bin < NC) reinterpret_cast<float2*>(obase)[N - bin] = make_float2(x.x, -x.y);
    } else {
        float v = cpow_mag(x, ep.power);
        if (ep.db) v = amp_to_db(v, ep.amin, ep.log10_ref);
        obase[bin] = v;
        if (!ep.onesided && bin > 0 && bin < NC) obase[N - bin] = v;
    }
}

// Loop-invariant per-lane constants: HOIST_* keeps them in registers for the whole kernel;
// otherwise the lane index is laundered through an empty asm each frame so the compiler cannot
// hoist the (L1-resident) table loads out of the frame loop and blow the register budget.
template <int NC, int E, int MODE, bool HOIST_TW, bool HOIST_WIN, bool HOIST_PTW>
__global__ void __launch_bounds__(STFT_WAVES * 64)
stft_kernel(FrameGeom g, Tables tb, StftEpilogue ep) {
    using F = WaveFft<NC, E>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* smem = reinterpret_cast<cf*>(smem_raw);

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int sub = lane / F::LPF;
    const int t = lane % F::LPF;
    cf* lds = smem + (w * F::G + sub) * F::PADDED;

    cf tw[F::NTW];
    float2 win[HOIST_WIN ? F::E : 1];
    cf ptw[HOIST_PTW ? F::NPAIR : 1];
    if constexpr (HOIST_TW) F::load_twiddles(tw, tb.w_nc, t);
    if constexpr (HOIST_WIN) load_window_regs<F>(win, g, t);
    if constexpr (HOIST_PTW) {
#pragma unroll
        for (int i = 0; i < F::NPAIR; ++i) ptw[i] = tb.w_n[t + i * F::LPF];
    }

    const long long groups_per_row = (g.n_frames + F::G - 1) / F::G;
    const long long total = g.rows * groups_per_row;
    const long long chunk = (total + gridDim.x - 1) / gridDim.x;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long end = begin + chunk < total ? begin + chunk : total;
    const int nbins = ep.onesided ? NC + 1 : 2 * NC;
    const long long per_frame = (long long)nbins * (MODE == 0 ? 2 : 1);

    for (long long grp = begin + w; grp < end; grp += STFT_WAVES) {
        const long long row = grp / groups_per_row;
        const long long frame = (grp - row * groups_per_row) * F::G + sub;
        cf v[F::E];
        int tl = t;
        if constexpr (!(HOIST_TW && HOIST_WIN && HOIST_PTW)) asm volatile("" : "+v"(tl));
        if constexpr (!HOIST_TW) F::load_twiddles(tw, tb.w_nc, tl);
        load_frame<F, HOIST_WIN>(v, g, win, row, frame, HOIST_WIN ? t : tl);
        F::run(v, lds, tw, t);
        if (frame < g.n_frames) {
            float* obase = ep.out + (row * g.n_frames + frame) * per_frame;
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) {
                const int k = t + i * F::LPF;
                cf xa, xb;
                F::r2c_pair(lds, k, HOIST_PTW ? ptw[i] : tb.w_n[tl + i * F::LPF], xa, xb);
                emit_bin<NC, MODE>(ep, obase, k, xa, g.scale);
                emit_bin<NC, MODE>(ep, obase, NC - k, xb, g.scale);
            }
            if (t == 0) {
                cf xa, xb;
                F::r2c_pair(lds, NC / 2, make_float2(0.0f, -1.0f), xa, xb);
                emit_bin<NC, MODE>(ep, obase, NC / 2, xa, g.scale);
            }
        }
        wave_lds_fence();   // next frame's first-pass writes must follow these reads
    }
}

template <int NC, int E, int MODE>
static int launch_stft(const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, hipStream_t stream) {
    using F = WaveFft<NC, E>;
    const long long groups = g.rows * ((g.n_frames + F::G - 1) / F::G);
    const size_t lds_bytes = (size_t)STFT_WAVES * F::G * F::PADDED * sizeof(cf);
    int per_cu = (int)(160 * 1024 / lds_bytes);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    long long max_blocks = (long long)device_cu_count() * per_cu;
    long long want = (groups + STFT_WAVES - 1) / STFT_WAVES;
    long long blocks = want < max_blocks ? want : max_blocks;
    if (blocks < 1) blocks = 1;
    constexpr bool H = (E <= 16);
    auto kern = stft_kernel<NC, E, MODE, H, false, false>;
    static bool attr_set = false;
    if (!attr_set && lds_bytes > 64 * 1024) {
        TAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(STFT_WAVES * 64), lds_bytes, stream, g, tb, ep);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

template <int MODE>
static int dispatch_stft(int n_fft, const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, hipStream_t s) {
    switch (n_fft) {
        case 32: return launch_stft<16, 16, MODE>(g, tb, ep, s);
        case 64: return launch_stft<32, 16, MODE>(g, tb, ep, s);
        case 128: return launch_stft<64, 16, MODE>(g, tb, ep, s);
        case 256: return launch_stft<128, 16, MODE>(g, tb, ep, s);
        case 512: return launch_stft<256, 16, MODE>(g, tb, ep, s);
        case 1024: return launch_stft<512, 16, MODE>(g, tb, ep, s);
        case 2048: return launch_stft<1024, 16, MODE>(g, tb, ep, s);
        case 4096: return launch_stft<2048, 32, MODE>(g, tb, ep, s);
        default: return TAC_E_UNSUPPORTED;
    }
}

}  // namespace tac

extern "C" {

int tac_stft_f32(const float* wave, const float* window, const tac_stft_desc* d, float* out, void* stream) {
    using namespace tac;
    if (!out) return TAC_E_INVALID;
    FrameGeom g;
    int64_t T = 0;
    int rc = make_geometry(wave, window, d, &g, &T);
    if (rc != TAC_OK) return rc;
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    StftEpilogue ep{out, d->onesided ? 1 : 0, 0, 1.0f, 0, 0.0f, 0.0f};
    return dispatch_stft<0>(d->n_fft, g, tb, ep, (hipStream_t)stream);
}

int tac_spectrogram_f32(const float* wave, const float* window, const tac_stft_desc* d, float power, int db,
                        float db_ref, float db_amin, float* out, void* stream) {
    using namespace tac;
    if (!out) return TAC_E_INVALID;
    FrameGeom g;
    int64_t T = 0;
    int rc = make_geometry(wave, window, d, &g, &T);
    if (rc != TAC_OK) return rc;
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    StftEpilogue ep{out, d->onesided ? 1 : 0, 1, power, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f};
    return dispatch_stft<1>(d->n_fft, g, tb, ep, (hipStream_t)stream);
}

}  // extern "C"
