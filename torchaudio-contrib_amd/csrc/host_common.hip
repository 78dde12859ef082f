// host_common.hip — error strings, geometry validation, twiddle-table cache.
#include "host_common.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

namespace tac {

thread_local int g_last_hip_error = 0;
thread_local char g_last_route[192] = "";
thread_local unsigned long long* g_clock_probe = nullptr;
thread_local int g_clock_probe_pairs = 0;

void set_last_route(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_route, sizeof(g_last_route), fmt, ap);
    va_end(ap);
}

#ifndef TAC_FFT_PIPE_DEFAULT_MFMA
#define TAC_FFT_PIPE_DEFAULT_MFMA 0   // 1: the fft_length-2048 kernels run their transform on the matrix pipe unless told otherwise
#endif
std::atomic<int> g_fft_pipe{-1};
bool fft_pipe_mfma() {
    const int m = g_fft_pipe.load(std::memory_order_relaxed);
    if (m >= 0) return m == 1;
    static const int env = [] { const char* e = getenv("TAC_FFT_PIPE"); return !e ? -1 : (e[0] == 'm' ? 1 : 0); }();
    return env >= 0 ? env == 1 : TAC_FFT_PIPE_DEFAULT_MFMA != 0;
}

int device_cu_count() {
    static thread_local int cached_dev = -1, cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            cached_cus = cus;
        cached_dev = dev;
    }
    return cached_cus;
}

hipError_t allow_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;              // (kernel, device) -> bytes granted
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(kernel, dev);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[key] = bytes;
    return e;
}

int get_tables(int n_fft, Tables* out) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, Tables> cache;
    int dev = 0;
    TAC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(n_fft, dev);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = it->second;
        return TAC_OK;
    }
    if (!is_pow2(n_fft)) {                      // sizes with their own tables (stft_n400.hip)
        *out = Tables{nullptr, nullptr, nullptr};
        return TAC_OK;
    }
    const int nc = n_fft / 2;
    const int n_post = nc / 2 + 1;
    const size_t img_cf = n_fft == 2048 ? (size_t)(S3_IMG_TW1_F4 + S3_IMG_PTW_F4) * 2 : 0;
    std::vector<cf> host((size_t)nc + n_post + img_cf);
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < nc; ++k) {
        double a = -two_pi * (double)k / (double)nc;
        host[k] = mkc((float)std::cos(a), (float)std::sin(a));
    }
    for (int k = 0; k < n_post; ++k) {
        double a = -two_pi * (double)k / (double)n_fft;
        host[nc + k] = mkc((float)std::cos(a), (float)std::sin(a));
    }
    const size_t img_at = ((size_t)nc + n_post + 1) & ~(size_t)1;                 // 16-byte aligned
    if (img_cf) {
        host.resize(img_at + img_cf, mkc(0.0f, 0.0f));
        float* tw1 = reinterpret_cast<float*>(host.data() + img_at);
        for (int js = 0; js < 16; ++js)                                            // set js: W_NC^{js q (NC/256)}, q = 1..15; slot 15 = 1
            for (int q = 0; q < 16; ++q) {
                const cf wv = q ? host[(size_t)js * q * (nc / 256)] : mkc(1.0f, 0.0f);
                tw1[js * S3_IMG_TW_STRIDE + 2 * (q ? q - 1 : 15)] = wv.x;
                tw1[js * S3_IMG_TW_STRIDE + 2 * (q ? q - 1 : 15) + 1] = wv.y;
            }
        cf* ptw = host.data() + img_at + (size_t)S3_IMG_TW1_F4 * 2;
        for (int tt = 0; tt < 64; ++tt)
            for (int p = 0; p < 8; ++p) ptw[((p >> 1) * 64 + tt) * 2 + (p & 1)] = host[(size_t)nc + tt + p * 64];
    }
    cf* dptr = nullptr;
    TAC_HIP(hipMalloc((void**)&dptr, host.size() * sizeof(cf)));
    TAC_HIP(hipMemcpy(dptr, host.data(), host.size() * sizeof(cf), hipMemcpyHostToDevice));
    Tables t{dptr, dptr + nc, img_cf ? reinterpret_cast<const float*>(dptr + img_at) : nullptr};
    cache[key] = t;
    *out = t;
    return TAC_OK;
}

int make_geometry(const float* wave, const float* window, const tac_stft_desc* d, FrameGeom* g,
                  int64_t* n_frames, bool any_size) {
    if (!wave || !window || !d || !g) return TAC_E_INVALID;
    if (d->rows <= 0 || d->length <= 0 || d->hop <= 0 || d->n_fft <= 0) return TAC_E_INVALID;
    if (d->win_length <= 0 || d->win_length > d->n_fft) return TAC_E_INVALID;
    if (d->pad_mode < TAC_PAD_CONSTANT || d->pad_mode > TAC_PAD_CIRCULAR) return TAC_E_INVALID;
    if (d->row_stride < d->length) return TAC_E_INVALID;
    if (d->length >= 0x7fffffffLL - 2 * (int64_t)d->n_fft) return TAC_E_UNSUPPORTED;   // 32-bit sample indices in-kernel
    if (!any_size && (!is_pow2(d->n_fft) || d->n_fft < 32 || d->n_fft > 4096) && d->n_fft != 400) return TAC_E_UNSUPPORTED;
    const int pad = d->center ? d->n_fft / 2 : 0;
    if (pad > 0) {
        // torch's reflect pad needs pad < L, circular needs pad <= L (functional.py:99-107 -> F.pad)
        if (d->pad_mode == TAC_PAD_REFLECT && pad >= d->length) return TAC_E_SHORT_INPUT;
        if (d->pad_mode == TAC_PAD_CIRCULAR && pad > d->length) return TAC_E_SHORT_INPUT;
    }
    const int64_t T = tac_num_frames(d->length, d->n_fft, d->hop, d->center);
    if (T <= 0) return TAC_E_SHORT_INPUT;
    g->wave = wave;
    g->row_stride = d->row_stride;
    g->length = d->length;
    g->window = window;
    g->win_length = d->win_length;
    g->win_offset = (d->n_fft - d->win_length) / 2;
    g->hop = d->hop;
    g->center_pad = pad;
    g->pad_mode = d->pad_mode;
    g->vec2_ok = ((d->hop & 1) == 0) && ((pad & 1) == 0) && ((d->row_stride & 1) == 0) &&
                 ((reinterpret_cast<uintptr_t>(wave) & 7u) == 0);
    g->vec4_ok = g->vec2_ok && ((d->hop & 3) == 0) && ((pad & 3) == 0) && ((d->row_stride & 3) == 0) &&
                 ((reinterpret_cast<uintptr_t>(wave) & 15u) == 0);
    g->n_frames = T;
    g->rows = d->rows;
    g->scale = d->normalized ? (float)(1.0 / std::sqrt((double)d->n_fft)) : 1.0f;
    *n_frames = T;
    return TAC_OK;
}

}  // namespace tac

extern "C" {

const char* tac_strerror(int code) {
    switch (code) {
        case TAC_OK: return "ok";
        case TAC_E_INVALID: return "invalid argument";
        case TAC_E_UNSUPPORTED: return "unsupported configuration for the HIP path";
        case TAC_E_SHORT_INPUT: return "input too short for the requested n_fft / padding";
        case TAC_E_LAUNCH: return "HIP runtime error";
        default: return "unknown error";
    }
}

int tac_last_hip_error(void) { return tac::g_last_hip_error; }

// 4: round 5 — tac_mulaw_encode_f64_i64 / tac_mulaw_decode_f64; tac_melbank_pack no longer takes the piece layout
// 3: round 4 — tac_last_route / tac_debug_clock_probe; the float64 entry points and the coded-input / fused-backward
//    launchers added during round 3 are counted from here as well (a library older than the binding fails its version check
//    in _native.lib() instead of at the first missing symbol)
// 5: round 6 — tac_set_fft_pipe
int tac_abi_version(void) { return 5; }

int tac_set_fft_pipe(int mode) {
    if (mode < -1 || mode > 1) return TAC_E_INVALID;
    return tac::g_fft_pipe.exchange(mode);
}

const char* tac_last_route(void) { return tac::g_last_route; }

int tac_debug_clock_probe(uint64_t* buf, int32_t capacity_pairs) {
    if (buf && capacity_pairs <= 0) return TAC_E_INVALID;
    tac::g_clock_probe = reinterpret_cast<unsigned long long*>(buf);
    tac::g_clock_probe_pairs = buf ? capacity_pairs : 0;
    return TAC_OK;
}

int64_t tac_num_frames(int64_t L, int n_fft, int hop, int center) {
    if (L <= 0 || n_fft <= 0 || hop <= 0) return 0;
    const int64_t padded = L + (center ? 2 * (int64_t)(n_fft / 2) : 0);
    if (padded < n_fft) return 0;
    return 1 + (padded - n_fft) / hop;
}

int tac_num_bins(int n_fft, int onesided) { return n_fft <= 0 ? 0 : (onesided ? n_fft / 2 + 1 : n_fft); }

}  // extern "C"
