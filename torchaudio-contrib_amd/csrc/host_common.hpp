// host_common.hpp — host-side helpers shared by the C-ABI launchers of libtac_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/tac_amd.h"
#include "fft_core.hpp"

namespace tac {

// epilogue selection of the STFT-family kernels (stft_kernels.hip, stft_n4096.hip)
struct StftEpilogue {
    float* out;
    int onesided;
    int mode;        // 0 complex, 1 magnitude^power
    float power;
    int db;
    float amin;
    float log10_ref;
};

#if defined(__HIPCC__)
// Real-valued row epilogues of the pipelined STFT kernels: MODE 1 = |X|^2, 2 = |X|, 3 = |X|^2 in dB, 4 = |X| in dB
// (amplitude_to_db squares its input, functional.py:291-296); 0 = complex rows.  |X| uses the hardware square root
// (1 ulp): the correctly rounded sequence costs ~10 instructions per bin.
template <int MODE>
__device__ __forceinline__ float spectral_row_value(float norm2, const StftEpilogue& ep) {
    float v = (MODE == 2 || MODE == 4) ? __builtin_amdgcn_sqrtf(norm2) : norm2;
    if constexpr (MODE >= 3) v = amp_to_db(v, ep.amin, ep.log10_ref);
    return v;
}
#endif


// per-bin entry of the band-sparse filterbank adjoint (built by fb_adjoint_pack_kernel, backward.hip): grad_spec[bin] =
// w0 grad_mel[b0] + w1 grad_mel[b1]
struct AdjEntry { float w0, w1; int b0, b1; };

// sample formats of the fused kernels' frame loads (= TAC_SAMPLES_*): float32, int16 PCM, mu-law codes as uint8 / int64
enum { FMT_F32 = 0, FMT_I16 = 1, FMT_MULAW_U8 = 2, FMT_MULAW_I64 = 3 };

extern thread_local int g_last_hip_error;
// diagnostics of include/tac_amd.h (11): the calling thread's last fused-chain kernel name and its clock-probe buffer
extern thread_local char g_last_route[192];
extern thread_local unsigned long long* g_clock_probe;
extern thread_local int g_clock_probe_pairs;
void set_last_route(const char* fmt, ...);

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return TAC_E_LAUNCH;
}
#define TAC_HIP(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return ::tac::hip_fail(_e); \
    } while (0)

struct Tables {
    const cf* w_nc;   // exp(-2*pi*i*k/NC), k < NC          (NC = n_fft/2)
    const cf* w_n;    // exp(-2*pi*i*k/N),  k <= NC/2
    // fft_length 2048 only (else null): the pass-1 twiddle sets and the R2C twiddles of the one-frame-per-wave kernels
    // (melspec_stream3.hpp, stft_stream3.hpp), already in their LDS layout — S3_IMG_TW1_F4 16-byte chunks of
    // [16 sets][ST_TW_STRIDE floats] followed by S3_IMG_PTW_F4 chunks of [pair >> 1][lane][pair & 1] — so that a workgroup's
    // set-up copies them with two 16-byte loads per thread instead of computing table indices per element
    const float* s3img;
};
constexpr int S3_IMG_TW_STRIDE = 36;                       // == ST_TW_STRIDE (melspec_stream.hpp)
constexpr int S3_IMG_TW1_F4 = 16 * S3_IMG_TW_STRIDE / 4;   // 144
constexpr int S3_IMG_PTW_F4 = 64 * 8 * 2 / 4;              // 256: 64 lanes x 8 pairs of complex values

// immutable per-(n_fft, device) twiddle tables; first use allocates + uploads (synchronous)
int get_tables(int n_fft, Tables* out);

int device_cu_count();

// tac_set_fft_pipe / TAC_FFT_PIPE: true when the fft_length-2048 kernels are to run their transform on the matrix pipe
extern std::atomic<int> g_fft_pipe;
bool fft_pipe_mfma();

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) — a process that uses several GPUs sets it
// on each of them; thread-safe.
hipError_t allow_dynamic_lds(const void* kernel, int bytes);

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Validates a descriptor the way torch.stft does and fills the device-side geometry.
// Returns TAC_OK or an error code.  T is written to *n_frames.
// any_size: framing only (overlap-add, window gradient) — accepts every fft_length, not just the FFT kernels' sizes
int make_geometry(const float* wave, const float* window, const tac_stft_desc* d, FrameGeom* g,
                  int64_t* n_frames, bool any_size = false);

}  // namespace tac
