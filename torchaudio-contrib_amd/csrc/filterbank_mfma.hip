// filterbank_mfma.hip — functional.apply_filterbank (functional.py:172-184) as an fp32 MFMA GEMM.
//
// out[r][t][m] = sum_f spec[r][f][t] * fb[f][m] for arbitrary (dense / random) banks and arbitrary spec strides (the
// reference hands over both (F,T)-contiguous tensors and the (T,F)-major strided views torch.stft produces).  The
// same kernel is the STFT for fft_length values the FFT kernels do not cover (non powers of two such as 400, and
// 4096 < N <= 8192): there `spec` is the padded waveform read as overlapping frames (stride_t = hop, stride_f = 1)
// and `fb` the windowed DFT matrix.  Exact f32 on v_mfma_f32_32x32x2_f32 — bf16 would break the 1e-4 parity budget
// and gfx950 has no xf32.
//
// Workgroup = 4 waves = 128 frames x 128 columns, each wave a 64 x 64 quadrant (2 x 2 MFMA tiles of 32 x 32, 64
// accumulator registers).  K is walked in 32-deep chunks through LDS: A^T as [32][132] (k-major, so a wave's operand
// read is 32 consecutive frames — conflict-free — and the global->LDS writes are too), B as [32][132] rows written
// 16 bytes at a time.  The next chunk's 8 x 16-byte global loads per thread are issued before the current chunk's
// 64 MFMAs per wave.  With a plan (tac_filterbank_plan) the K range of a column tile shrinks to the union of its
// 16-band tiles' supports (block-sparse banks); plan == NULL runs dense.
#include "host_common.hpp"

namespace tac {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float gm_f4 __attribute__((ext_vector_type(4)));

constexpr int GM_TN = 128, GM_KC = 32, GM_LD = 132;

// TM = frames per workgroup: 128 (each wave a 64 x 64 quadrant) or 64 (each wave 32 x 64) — the smaller tile halves
// the work quantum when a problem is only a few tiles per CU (cfg-2: 2.4 tiles of 128 per CU, i.e. 3 rounds for 2.4)
template <int GM_TM>
__global__ void __launch_bounds__(256, 2)
gemm_fb_kernel(const float* __restrict__ spec, long long stride_r, long long stride_f, long long stride_t, int n_freqs,
               long long n_frames, long long frame_tiles, int col_tiles, const float* __restrict__ fb,
               const int* __restrict__ plan, int n_mels, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float at_lds[GM_KC * GM_LD];       // A^T chunk: [k][frame]
    __shared__ __attribute__((aligned(16))) float b_lds[GM_KC * GM_LD];        // B chunk:   [k][column]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int MI = GM_TM / 64;                                              // 32-row MFMA tiles per wave along M
    const int wm = w >> 1, wn = w & 1;                                          // the wave's (32*MI) x 64 quadrant
    long long bid = blockIdx.x;
    const int ct = (int)(bid % col_tiles);
    bid /= col_tiles;
    const long long row = bid / frame_tiles;
    const long long f0 = (bid - row * frame_tiles) * GM_TM;
    const int c0 = ct * GM_TN;
    const float* srow = spec + row * stride_r;

    // K range of this column tile (block-sparse banks), rounded to whole chunks
    int klo = 0, khi = n_freqs;
    if (plan) {
        klo = n_freqs;
        khi = 0;
        const int n_band_tiles = (n_mels + 15) / 16;
        for (int j = 0; j < GM_TN / 16; ++j) {
            const int bt = ct * (GM_TN / 16) + j;
            if (bt < n_band_tiles) {
                const int lo = plan[2 * bt], hi = plan[2 * bt + 1];
                if (hi > lo) { klo = lo < klo ? lo : klo; khi = hi > khi ? hi : khi; }
            }
        }
        klo = (klo / GM_KC) * GM_KC;
    }

    // global -> register staging of one chunk: A as 4 x (4 consecutive k of one frame), B as 4 x (4 consecutive columns)
    constexpr int AJ = GM_TM / 32;                                              // 16-byte A loads per thread per chunk
    const int a_i = tid & (GM_TM - 1), a_kq = tid / GM_TM;                      // frame within tile, k slice (4*AJ each)
    const int b_j4 = tid & 31, b_k = tid >> 5;                                  // column group, k row (0..7, +8 per step)
    const long long a_frame = f0 + a_i;
    const bool a_live = a_frame < n_frames;
    const float* a_src = srow + (a_live ? a_frame : n_frames - 1) * stride_t;
    gm_f4 ra[AJ], rb[4];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int k = kc + a_kq * (4 * AJ) + 4 * j;
            gm_f4 v;
            if (stride_f == 1 && k + 4 <= n_freqs) {
                v = *reinterpret_cast<const gm_f4*>(a_src + k);                  // dword alignment is all a global load needs
            } else {
                v.x = k < n_freqs ? a_src[(long long)k * stride_f] : 0.0f;
                v.y = k + 1 < n_freqs ? a_src[(long long)(k + 1) * stride_f] : 0.0f;
                v.z = k + 2 < n_freqs ? a_src[(long long)(k + 2) * stride_f] : 0.0f;
                v.w = k + 3 < n_freqs ? a_src[(long long)(k + 3) * stride_f] : 0.0f;
            }
            ra[j] = a_live ? v : gm_f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kc + b_k + 8 * j;
            const int col = c0 + 4 * b_j4;
            gm_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (k < n_freqs) {
                const float* bp = fb + (long long)k * n_mels + col;
                if (col + 4 <= n_mels) {
                    v = *reinterpret_cast<const gm_f4*>(bp);
                } else {
                    v.x = col < n_mels ? bp[0] : 0.0f;
                    v.y = col + 1 < n_mels ? bp[1] : 0.0f;
                    v.z = col + 2 < n_mels ? bp[2] : 0.0f;
                    v.w = col + 3 < n_mels ? bp[3] : 0.0f;
                }
            }
            rb[j] = v;
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int k = a_kq * (4 * AJ) + 4 * j;
            at_lds[(k + 0) * GM_LD + a_i] = ra[j].x;
            at_lds[(k + 1) * GM_LD + a_i] = ra[j].y;
            at_lds[(k + 2) * GM_LD + a_i] = ra[j].z;
            at_lds[(k + 3) * GM_LD + a_i] = ra[j].w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<gm_f4*>(b_lds + (b_k + 8 * j) * GM_LD + 4 * b_j4) = rb[j];
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.0f;

    const int li = lane & 31, lk = lane >> 5;
    if (klo < khi) fetch(klo);
    for (int kc = klo; kc < khi; kc += GM_KC) {
        __syncthreads();                                   // the previous chunk's operand reads are done
        deposit();
        __syncthreads();
        if (kc + GM_KC < khi) fetch(kc + GM_KC);           // in flight during this chunk's MFMAs
#pragma unroll
        for (int k2 = 0; k2 < GM_KC / 2; ++k2) {
            const int k = 2 * k2 + lk;
            const float b0 = b_lds[k * GM_LD + wn * 64 + li], b1 = b_lds[k * GM_LD + wn * 64 + 32 + li];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const float a = at_lds[k * GM_LD + wm * (32 * MI) + 32 * mi + li];
                acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[mi][0], 0, 0, 0);
                acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[mi][1], 0, 0, 0);
            }
        }
    }
    // D[i][j] of a 32 x 32 tile: j = lane % 32, i = 8*(r / 4) + 4*(lane / 32) + r % 4
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int col = c0 + wn * 64 + nj * 32 + li;
            if (col < n_mels) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long frame = f0 + wm * (32 * MI) + mi * 32 + 8 * (r >> 2) + 4 * lk + (r & 3);
                    if (frame < n_frames) out[(row * n_frames + frame) * n_mels + col] = acc[mi][nj][r];
                }
            }
        }
}

}  // namespace tac

extern "C" {

int tac_apply_filterbank_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                             int64_t stride_f, int64_t stride_t, const float* fb, const int32_t* fb_plan,
                             int32_t n_mels, float* out, void* stream) {
    using namespace tac;
    if (!spec || !fb || !out) return TAC_E_INVALID;
    if (rows <= 0 || n_freqs <= 0 || n_frames <= 0 || n_mels <= 0) return TAC_E_INVALID;
    // rows whose frames continue each other in memory (the frame-major tensors the kernels here write) are one long
    // run of frames: no partly filled 128-frame tile at the end of every row (313 frames per row: 18 % fewer tiles)
    if (rows > 1 && stride_r == n_frames * stride_t) {
        n_frames *= rows;
        rows = 1;
    }
    const int col_tiles = (n_mels + GM_TN - 1) / GM_TN;
    const long long tiles128 = rows * ((n_frames + 127) / 128) * col_tiles;
    const int tm = tiles128 < 8LL * device_cu_count() ? 64 : 128;               // few tiles per CU: halve the quantum
    const long long frame_tiles = (n_frames + tm - 1) / tm;
    const long long blocks = rows * frame_tiles * col_tiles;
    if (blocks > 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    if (tm == 64)
        hipLaunchKernelGGL(gemm_fb_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, spec,
                           (long long)stride_r, (long long)stride_f, (long long)stride_t, n_freqs, (long long)n_frames,
                           frame_tiles, col_tiles, fb, fb_plan, n_mels, out);
    else
        hipLaunchKernelGGL(gemm_fb_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, spec,
                           (long long)stride_r, (long long)stride_f, (long long)stride_t, n_freqs, (long long)n_frames,
                           frame_tiles, col_tiles, fb, fb_plan, n_mels, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // extern "C"
