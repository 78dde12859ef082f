// filterbank_mfma.hip — functional.apply_filterbank (functional.py:172-184) as an fp32 MFMA GEMM.
//
// out[r][t][m] = sum_f spec[r][f][t] * fb[f][m] for arbitrary (dense / random) banks and arbitrary spec strides (the
// reference hands over both (F,T)-contiguous tensors and the (T,F)-major strided views torch.stft produces).  The
// same kernel is the STFT for fft_length values the FFT kernels do not cover (non powers of two such as 400, and
// 4096 < N <= 8192): there `spec` is the padded waveform read as overlapping frames (stride_t = hop, stride_f = 1)
// and `fb` the windowed DFT matrix.  Exact f32 on v_mfma_f32_32x32x2_f32 — bf16 would break the 1e-4 parity budget
// and gfx950 has no xf32.
//
// Workgroup = 8 waves = 64 frames x 128 columns, each wave ONE 32 x 32 MFMA tile (16 accumulator registers).  Round 6: until then
// four waves held 2 x 2 tiles each on a 128 x 128 (or 64 x 128) workgroup tile, two workgroups per CU: cfg-2's 80 128 frames were
// 626 (1 252) tiles for 512 slots — 1.22 (2.45) rounds paid as 2 (3) — and every chunk cost two barriers around a single LDS buffer:
// matrix pipe 63 % busy, 57 % of the f32 MFMA peak (profiles/r05/pmc_fb.json).  Now the unit of work a SIMD sees is one 32 x 32 tile
// over the whole K range (80 128 x 128 outputs = 10 016 tiles on 1 024 SIMDs: 9.78 rounds paid as 10), three workgroups = six waves
// per SIMD keep the matrix pipe fed across each other's barriers, and K is walked in 32-deep chunks through TWO LDS buffers: the next
// chunk's global loads are issued before the current chunk's 16 MFMAs per wave and deposited into the other buffer after them — one
// barrier per chunk.  A^T as [32][68] (k-major: a wave's operand read is 32 consecutive frames, the global -> LDS writes 64), B as
// [32][132] rows written 16 bytes at a time.  With a plan (tac_filterbank_plan) the K range of a column tile shrinks to the union
// of its 16-band tiles' supports (block-sparse banks); plan == NULL runs dense.
#include "host_common.hpp"

namespace tac {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float gm_f4 __attribute__((ext_vector_type(4)));

constexpr int GM_TM = 64, GM_TN = 128, GM_KC = 32, GM_LDA = 66, GM_LDB = 132, GM_THREADS = 512;

// element i of (v.x, v.y, v.z, v.w, 0, 0, 0): the tail of a row read 16 bytes at a time from a clamped address
__device__ __forceinline__ float gm_pick(gm_f4 v, int i) {
    const float lo = (i & 1) ? v.y : v.x, hi = (i & 1) ? v.w : v.z;
    return i < 4 ? ((i & 2) ? hi : lo) : 0.0f;
}

// VA: spec is read 16 bytes at a time along the frequency axis (stride_f == 1, n_freqs >= 4); VB: the bank's rows likewise
// (n_mels a multiple of four).  The requests of a chunk are UNCONDITIONAL loads from clamped addresses — no branch, no select on a
// loaded value before the deposit (round 5's per-element conditions made the compiler wait for every load right where it was
// issued: the prefetch never ran ahead, and every chunk paid a memory round trip) — and what lies outside the matrices is zeroed
// when the chunk is deposited.
template <bool VA, bool VB>
__global__ void __launch_bounds__(GM_THREADS, 6)
gemm_fb_kernel(const float* __restrict__ spec, long long stride_r, long long stride_f, long long stride_t, int n_freqs,
               long long n_frames, long long frame_tiles, int col_tiles, const float* __restrict__ fb,
               const int* __restrict__ plan, int n_mels, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float at_lds[2][GM_KC * GM_LDA];   // A^T chunks: [k][frame]
    __shared__ __attribute__((aligned(16))) float b_lds[2][GM_KC * GM_LDB];    // B chunks:   [k][column]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;                                          // the wave's 32 x 32 tile of the 64 x 128
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

    // one chunk per thread: A = 4 consecutive k of one frame — eight consecutive lanes cover the 128 bytes of a frame's chunk
    // (LDA = 66: the transposed LDS writes of a wave, 8 frames x 32 k, hit 64 different banks) —, B = 2 x (4 consecutive columns)
    const int a_i = tid >> 3, a_kq = tid & 7;                                   // frame within tile, k slice (4 each)
    const int b_j4 = tid & 31, b_k = tid >> 5;                                  // column group, k row (0..15, +16)
    const long long a_frame = f0 + a_i;
    const bool a_live = a_frame < n_frames;
    const float* a_src = srow + (a_live ? a_frame : n_frames - 1) * stride_t;
    const int col = c0 + 4 * b_j4;
    const bool b_in = col < n_mels;
    const float* b_src = fb + (b_in ? col : 0);
    struct Stage { gm_f4 a, b0, b1; };
    auto fetch = [&](int kc) {
        Stage st;
        const int k = kc + 4 * a_kq;
        if constexpr (VA) {
            const int kk = k < n_freqs - 4 ? k : n_freqs - 4;
            st.a = *reinterpret_cast<const gm_f4*>(a_src + kk);                 // dword alignment is all a global load needs
        } else {
            const int last = n_freqs - 1;
            st.a.x = a_src[(long long)(k < last ? k : last) * stride_f];
            st.a.y = a_src[(long long)(k + 1 < last ? k + 1 : last) * stride_f];
            st.a.z = a_src[(long long)(k + 2 < last ? k + 2 : last) * stride_f];
            st.a.w = a_src[(long long)(k + 3 < last ? k + 3 : last) * stride_f];
        }
        const int r0 = kc + b_k, r1 = r0 + 16, lastr = n_freqs - 1;
        const float* p0 = b_src + (long long)(r0 < lastr ? r0 : lastr) * n_mels;
        const float* p1 = b_src + (long long)(r1 < lastr ? r1 : lastr) * n_mels;
        if constexpr (VB) {
            st.b0 = *reinterpret_cast<const gm_f4*>(p0);
            st.b1 = *reinterpret_cast<const gm_f4*>(p1);
        } else {
            const int c1 = col + 1 < n_mels ? 1 : 0, c2 = col + 2 < n_mels ? 2 : 0, c3 = col + 3 < n_mels ? 3 : 0;   // (clamped: masked at the deposit)
            st.b0.x = p0[0]; st.b0.y = p0[c1]; st.b0.z = p0[c2]; st.b0.w = p0[c3];
            st.b1.x = p1[0]; st.b1.y = p1[c1]; st.b1.z = p1[c2]; st.b1.w = p1[c3];
        }
        return st;
    };
    auto deposit = [&](int buf, const Stage& st, int kc) {
        float* at = at_lds[buf];
        const int k = kc + 4 * a_kq, kl = 4 * a_kq;
        gm_f4 va;
        if constexpr (VA) {
            const int sh = k < n_freqs - 4 ? 0 : k - (n_freqs - 4);             // (> 0 only in the row's last chunk)
            va.x = gm_pick(st.a, sh); va.y = gm_pick(st.a, sh + 1); va.z = gm_pick(st.a, sh + 2); va.w = gm_pick(st.a, sh + 3);
        } else {
            va.x = k < n_freqs ? st.a.x : 0.0f; va.y = k + 1 < n_freqs ? st.a.y : 0.0f;
            va.z = k + 2 < n_freqs ? st.a.z : 0.0f; va.w = k + 3 < n_freqs ? st.a.w : 0.0f;
        }
        at[(kl + 0) * GM_LDA + a_i] = a_live ? va.x : 0.0f;
        at[(kl + 1) * GM_LDA + a_i] = a_live ? va.y : 0.0f;
        at[(kl + 2) * GM_LDA + a_i] = a_live ? va.z : 0.0f;
        at[(kl + 3) * GM_LDA + a_i] = a_live ? va.w : 0.0f;
        const bool in0 = b_in && kc + b_k < n_freqs, in1 = b_in && kc + b_k + 16 < n_freqs;
        gm_f4 v0, v1;
        v0.x = in0 ? st.b0.x : 0.0f; v0.y = in0 && col + 1 < n_mels ? st.b0.y : 0.0f;
        v0.z = in0 && col + 2 < n_mels ? st.b0.z : 0.0f; v0.w = in0 && col + 3 < n_mels ? st.b0.w : 0.0f;
        v1.x = in1 ? st.b1.x : 0.0f; v1.y = in1 && col + 1 < n_mels ? st.b1.y : 0.0f;
        v1.z = in1 && col + 2 < n_mels ? st.b1.z : 0.0f; v1.w = in1 && col + 3 < n_mels ? st.b1.w : 0.0f;
        *reinterpret_cast<gm_f4*>(b_lds[buf] + b_k * GM_LDB + 4 * b_j4) = v0;
        *reinterpret_cast<gm_f4*>(b_lds[buf] + (b_k + 16) * GM_LDB + 4 * b_j4) = v1;
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int li = lane & 31, lk = lane >> 5;
    auto compute = [&](int b) {
        const float* at = at_lds[b] + wm * 32 + li;
        const float* bt = b_lds[b] + wn * 32 + li;
#pragma unroll
        for (int k2 = 0; k2 < GM_KC / 2; ++k2) {
            const int k = 2 * k2 + lk;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[k * GM_LDA], bt[k * GM_LDB], acc, 0, 0, 0);
        }
    };
    // two register stages, the loop unrolled by two: the loads of chunk c + 2 go out while chunk c is multiplied and chunk c + 1
    // (requested a whole iteration earlier) is deposited — a global load has two chunks' time to land; one barrier per chunk.
    // The requests are unconditional (past the end they repeat the last chunk): a request under a condition makes the compiler's
    // wait-count bookkeeping assume the worst of both paths and wait for everything in flight.
    const int nch = klo < khi ? (khi - klo + GM_KC - 1) / GM_KC : 0;
    auto chunk_at = [&](int c) { return klo + (c < nch ? c : nch - 1) * GM_KC; };
    if (nch > 0) {
        Stage s0 = fetch(chunk_at(0));
        Stage s1 = fetch(chunk_at(1));
        __builtin_amdgcn_sched_barrier(0);
        deposit(0, s0, chunk_at(0));
        __syncthreads();
        for (int c = 0; c + 2 <= nch; c += 2) {
            s0 = fetch(chunk_at(c + 2));
            __builtin_amdgcn_sched_barrier(0);             // (the scheduler would sink the requests to where their registers are used)
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            deposit(1, s1, chunk_at(c + 1));
            __syncthreads();
            s1 = fetch(chunk_at(c + 3));
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            deposit(0, s0, chunk_at(c + 2));               // (unconditional as well — under a condition the compiler sinks the request
            __syncthreads();                               //  itself into the branch; past the end the buffer is not read again)
        }
        if (nch & 1) compute(0);
    }
    // D[i][j] of a 32 x 32 tile: j = lane % 32, i = 8*(r / 4) + 4*(lane / 32) + r % 4
    const int ocol = c0 + wn * 32 + li;
    if (ocol < n_mels) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long frame = f0 + wm * 32 + 8 * (r >> 2) + 4 * lk + (r & 3);
            if (frame < n_frames) out[(row * n_frames + frame) * n_mels + ocol] = acc[r];
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
    const long long frame_tiles = (n_frames + GM_TM - 1) / GM_TM;
    const long long blocks = rows * frame_tiles * col_tiles;
    if (blocks > 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const bool va = stride_f == 1 && n_freqs >= 4, vb = (n_mels & 3) == 0;
    auto kern = va ? (vb ? gemm_fb_kernel<true, true> : gemm_fb_kernel<true, false>)
                   : (vb ? gemm_fb_kernel<false, true> : gemm_fb_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(GM_THREADS), 0, (hipStream_t)stream, spec,
                       (long long)stride_r, (long long)stride_f, (long long)stride_t, n_freqs, (long long)n_frames,
                       frame_tiles, col_tiles, fb, fb_plan, n_mels, out);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // extern "C"
