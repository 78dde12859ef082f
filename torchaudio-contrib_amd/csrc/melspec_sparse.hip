// melspec_sparse.hip — the fused Melspectrogram chain with a BAND-SPARSE VALU filterbank contraction.
//
// Same reference chain and phase A as melspec_fused.hip (layers.py:307-381), different phase B.  A triangular
// mel bank has ~16 non-zero weights per band out of 1025 (1.5 %); the MFMA formulation executes 16x16x4 tiles
// over each 16-band tile's bin range — 38 400 flop per frame for 4 042 useful ones, ~6 000 cycles per 16-frame
// tile — and needs partial tiles + a reduction pass because the K ranges are split over waves.  Here every
// (frame, band) output is one thread's private dot product over the band's CONTIGUOUS bin run:
//   * 16 lanes = the 16 frames of the tile share one band at a time (weight reads are LDS broadcasts, the
//     power-row reads hit 16 distinct banks because the row stride is 2 mod 32);
//   * bands are dealt to the lane groups longest-first (LPT) by tac_melbank_pack, so all groups of a wave
//     run inner loops of similar length at the same time and finish together;
//   * weights live in LDS as 16-byte aligned, zero-padded runs: one ds_read_b128 feeds four FMAs.
// No MFMA, no partial tiles, no weight registers: the 16 KB of partial slots become the weight + output tile
// storage and the FFT phase gets 40 more registers.  Banks that are not band-sparse enough for the LDS budget
// return TAC_E_UNSUPPORTED and take the MFMA kernels instead.
#include "mel_common.hpp"
#include "mel_lanes.hpp"

#include <algorithm>
#include <cstdlib>
#include <functional>
#include <atomic>
#include <vector>


#include "sparse_phase.hpp"
#include "melspec_stream.hpp"
#include "melspec_stream3.hpp"
#include "melspec_mfma.hpp"
#include "mel_pieces.hpp"
#include "lane_placement.hpp"

namespace tac {

// stft_n400.hip: the fused chain for fft_length 400
int launch_n400_mel(const FrameGeom& g, float power, const float* wpack, const int* desc, const int32_t* info_host,
                    int n_mels, int db, float amin, float log10_ref, float* out, hipStream_t stream, int fmt = FMT_F32,
                    const void* samples = nullptr, const float* lut = nullptr);
int pack_n400(const std::vector<float>& h, int n_freqs, int n_mels, float* wpack, int wpack_cap, int32_t* desc,
              int desc_cap, int32_t* info_host, hipStream_t stream, bool to_host = false);
// stft_small.hip: the same form for fft_length 512 / 1024 (the three-phase kernel below stays the fallback)
int launch_small_mel_entry(int n_fft, const FrameGeom& g, const Tables& tb, float power, const float* wpack, const int* desc,
                           const int32_t* info_host, int n_mels, int db, float amin, float log10_ref, float* out,
                           hipStream_t stream, int fmt = FMT_F32, const void* samples = nullptr, const float* lut = nullptr);
int pack_small(int n_fft, const std::vector<float>& h, int n_freqs, int n_mels, float* wpack, int wpack_cap, int32_t* desc,
               int desc_cap, int32_t* info_host, hipStream_t stream, bool to_host = false);
// stft_n4096.hip: fft_length 4096 (the twelve-wave form of its real-valued rows + the contraction, stft_n4096_s3.hpp)
int launch_n4096_mel(const FrameGeom& g, float power, const float* wpack, const int32_t* desc, const int32_t* info_host, int n_mels,
                     int db, float amin, float log10_ref, float* out, hipStream_t stream);
int pack_n4096_mel(const std::vector<float>& h, int n_freqs, int n_mels, float* wpack, int wpack_cap, int32_t* desc, int desc_cap,
                   int32_t* info_host, hipStream_t stream, bool to_host = false);

constexpr int SP_TILE = 16;
constexpr int SP_MAX_W = 3072;               // floats of packed weights that may live in LDS (12 KB)

struct SparseArgs {
    const float* wpack;    // device, wtot floats
    const int* desc;       // device, groups x dstride ints: {nb,0,0,0} then nb x {band, first bin, n8, weight offset}
    int wtot;              // multiple of 4
    int dstride;           // multiple of 4
    int n_mels;
    int db;
    float amin;
    float log10_ref;
    float* out;            // [rows][T][M]
    int out_vec4;          // n_mels % 4 == 0 and out 16-byte aligned: phase C moves four bands per lane
};

__host__ __device__ inline int sparse_ostr(int n_mels, int out_vec4) { return out_vec4 ? n_mels + 4 : (n_mels | 1); }

// phase B lives in sparse_phase.hpp (shared with the streaming kernel)

// phase C: dB epilogue + coalesced row stores of out[row][frame][0..M) by the NT threads that share the tile
template <int NT>
__device__ __forceinline__ void sparse_phase_c(const float* otile, int ostr, int th, int tile_frames, const SparseArgs& m,
                                               const FrameGeom& g, int row, long long f0) {
    if (m.out_vec4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int q4 = m.n_mels >> 2;
        const int c4_df = NT / q4, c4_db = NT % q4;
        int fo = th / q4, b4 = th % q4;
        for (int idx = th; idx < tile_frames * q4; idx += NT) {
            f4 v = *reinterpret_cast<const f4*>(otile + fo * ostr + 4 * b4);
            if (m.db) {
                v.x = amp_to_db(v.x, m.amin, m.log10_ref);
                v.y = amp_to_db(v.y, m.amin, m.log10_ref);
                v.z = amp_to_db(v.z, m.amin, m.log10_ref);
                v.w = amp_to_db(v.w, m.amin, m.log10_ref);
            }
            const long long frame = f0 + fo;
            if (frame < g.n_frames)
                *reinterpret_cast<f4*>(m.out + (row * g.n_frames + frame) * m.n_mels + 4 * b4) = v;
            b4 += c4_db;
            fo += c4_df;
            if (b4 >= q4) { b4 -= q4; ++fo; }
        }
    } else {
        const int c_df = NT / m.n_mels, c_dband = NT % m.n_mels;
        int fo = th / m.n_mels, band = th % m.n_mels;
        for (int idx = th; idx < tile_frames * m.n_mels; idx += NT) {
            float v = otile[fo * ostr + band];
            if (m.db) v = amp_to_db(v, m.amin, m.log10_ref);
            const long long frame = f0 + fo;
            if (frame < g.n_frames) m.out[(row * g.n_frames + frame) * m.n_mels + band] = v;
            band += c_dband;
            fo += c_df;
            if (band >= m.n_mels) { band -= m.n_mels; ++fo; }
        }
    }
}

template <int NC, int E, bool POW2>
__global__ void __launch_bounds__((MelCfg<NC, E, SP_TILE>::WAVES * 64), (MelCfg<NC, E, SP_TILE>::WAVES / 4))
melspec_sparse_kernel(FrameGeom g, Tables tb, SparseArgs m) {
    using C = MelCfg<NC, E, SP_TILE>;
    using F = typename C::F;
    constexpr int WAVES = C::WAVES, PROW = C::PROW, TILE = SP_TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* bufs = reinterpret_cast<cf*>(smem_raw);                                   // NBUF frame buffers
    float* wlds = reinterpret_cast<float*>(bufs + C::NBUF * F::PADDED);           // packed weights (16-byte aligned)
    // output-tile row stride: n_mels + 4 keeps rows 16-byte aligned for phase C's float4 reads (the 16 frames of a band
    // then write 2-way bank-conflicted, which ds_write_b32 absorbs); otherwise an odd stride, conflict-free columns
    const int ostr = sparse_ostr(m.n_mels, m.out_vec4);
    float* otile = wlds + m.wtot;                                                 // [TILE][ostr]
    int* dlds = reinterpret_cast<int*>(otile + ((TILE * ostr + 3) & ~3));         // [groups][dstride]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / F::LPF;
    const int t = lane % F::LPF;

    for (int i = tid; i < m.wtot; i += WAVES * 64) wlds[i] = m.wpack[i];
    for (int i = tid; i < WAVES * 4 * m.dstride; i += WAVES * 64) dlds[i] = m.desc[i];

    // register-resident for the kernel's lifetime: the window, the pass twiddles, and the R2C twiddles as ONE lane
    // register x compile-time W_32^i (hoisting all eight is 0.9 % faster but spills 28 B per lane)
    MelFftConsts<F, true, true> fftk;
    fftk.load(tb, g, t);
    __syncthreads();

    const int tiles_per_row = (int)((g.n_frames + TILE - 1) / TILE);
    const int total_tiles = (int)g.rows * tiles_per_row;
    const int chunk = (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total_tiles ? begin + chunk : total_tiles;

    // phase B identity of this thread: lane group (16 lanes) = one band at a time, lane in group = frame
    const int fr = tid & 15;
    const int* dg = dlds + (tid >> 4) * m.dstride;
    const float* prow = reinterpret_cast<const float*>(bufs) + fr * PROW;
    // phase C walks idx = tid, tid + T, ... over (frame, band) = (idx / M, idx % M) without dividing per element
    const int c_f0 = tid / m.n_mels, c_band0 = tid % m.n_mels;
    const int c_df = (WAVES * 64) / m.n_mels, c_dband = (WAVES * 64) % m.n_mels;
    const int q4 = m.n_mels >> 2;                        // float4 groups per output row (out_vec4 path)
    const int c4_f0 = q4 ? tid / q4 : 0, c4_b0 = q4 ? tid % q4 : 0;
    const int c4_df = q4 ? (WAVES * 64) / q4 : 0, c4_db = q4 ? (WAVES * 64) % q4 : 0;

    // every frame's raw samples are requested one frame ahead (mel_common.hpp); the first request opens the pipeline
    cf raw[F::E];
    bool pre_ok = false;
    if (begin < end) {
        const int r0 = begin / tiles_per_row;
        const long long fr0 = (long long)(begin - r0 * tiles_per_row) * TILE + (long long)w * C::GPW * F::G + sub;
        pre_ok = prefetch_frame_raw_x<F>(raw, g, r0, fr0, t);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the tile loop is entered with nothing in flight

    NoStamp st;
    for (int tile = begin; tile < end; ++tile) {
        const int row = tile / tiles_per_row;
        const long long f0 = (long long)(tile - row * tiles_per_row) * TILE;
        st.mark(0);

        // ---------------- phase A: FFT, then overwrite each frame buffer with its |X|^p row (mel_common.hpp)
        {
            const int nt = tile + 1;
            const int nr = nt / tiles_per_row;
            const long long nf0 = nt < end ? (long long)(nt - nr * tiles_per_row) * TILE : -1;
            mel_phase_a<C, POW2, true, decltype(st), true>(g, bufs, fftk, w, sub, t, row, f0, raw, &pre_ok, &st, nr, nf0);
        }
        __syncthreads();
        st.mark(1);                                        // barrier A (waiting for the slowest wave's FFTs)

        // ---------------- phase B: one private dot product per (frame, band)
        sparse_phase_b(dg, prow, wlds, otile + fr * ostr);
        st.mark(7);                                        // phase B
        __syncthreads();
        st.mark(11);                                       // barrier B

        // ---------------- phase C: dB epilogue + coalesced row stores of out[row][frame][0..M)
        sparse_phase_c<WAVES * 64>(otile, ostr, tid, TILE, m, g, row, f0);
        // the barrier after the next phase A orders these otile reads before the next phase-B writes
    }
}

// ---------------------------------------------------------------- standalone band-sparse filterbank
// functional.apply_filterbank (functional.py:172-184) for a frame-major spectrogram (bins of a frame contiguous: the
// layout every kernel of this library writes) and a band-sparse bank: the fused kernel with phase A replaced by
// "load 16 rows": an 8-wave workgroup holds 16 frames' rows in LDS, runs the same contraction (phase B) and epilogue
// (phase C).  The next tile's rows are requested into registers before this tile's contraction and written to LDS
// after it, so the kernel streams: spectrogram in (4·F bytes per frame), bands out.
constexpr int FBS_WAVES = 8, FBS_TILE = 16, FBS_CHUNKS = 9;       // 9 x 512 16-byte chunks per tile: up to 1152 bins per frame

__global__ void __launch_bounds__(FBS_WAVES * 64, 2)
fb_sparse_kernel(const float* __restrict__ spec, long long rows, int n_freqs, long long n_frames, long long stride_r,
                 long long stride_t, int prow_stride, SparseArgs m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* ptile = reinterpret_cast<float*>(smem_raw);                            // [16][prow_stride]
    float* wlds = ptile + FBS_TILE * prow_stride;
    const int ostr = sparse_ostr(m.n_mels, m.out_vec4);
    float* otile = wlds + m.wtot;
    int* dlds = reinterpret_cast<int*>(otile + ((FBS_TILE * ostr + 3) & ~3));
    const int tid = threadIdx.x;
    for (int i = tid; i < m.wtot; i += FBS_WAVES * 64) wlds[i] = m.wpack[i];
    for (int i = tid; i < FBS_WAVES * 4 * m.dstride; i += FBS_WAVES * 64) dlds[i] = m.desc[i];

    FrameGeom g{};                                        // phase C only needs the frame count
    g.n_frames = n_frames;
    const int tiles_per_row = (int)((n_frames + FBS_TILE - 1) / FBS_TILE);
    const int total_tiles = (int)rows * tiles_per_row;
    const int chunk = (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total_tiles ? begin + chunk : total_tiles;
    const int fr = tid & 15;
    const int* dg = dlds + (tid >> 4) * m.dstride;
    const float* prow = ptile + fr * prow_stride;

    // A tile's 16 rows are one contiguous span of 16*F floats when the frames are packed (stride_t == F, the layout
    // the spectrogram kernels write): it is fetched as 16-byte chunks (global loads only need dword alignment) and
    // scattered into the padded LDS rows; other frame strides fetch row by row.  Chunks past the end of the batch
    // row are clamped (and zeroed at deposit time).
    const bool packed = (stride_t == n_freqs);
    typedef float fbs_f4 __attribute__((ext_vector_type(4)));
    fbs_f4 nxt[FBS_CHUNKS];
    auto request = [&](int tile) {
        const int row = tile / tiles_per_row;
        const long long f0 = (long long)(tile - row * tiles_per_row) * FBS_TILE;
        const float* base = spec + row * stride_r;
        const long long row_floats = packed ? n_frames * n_freqs : 0;
#pragma unroll
        for (int j = 0; j < FBS_CHUNKS; ++j) {
            const int c = tid + FBS_WAVES * 64 * j;                      // chunk of the tile's span
            if (packed) {
                const long long e = f0 * n_freqs + 4LL * c;              // first element, relative to the batch row
                if (e + 4 <= row_floats) {
                    nxt[j] = *reinterpret_cast<const fbs_f4*>(base + e);
                } else {                                                 // the one chunk that straddles the row's end, and
                    fbs_f4 v;                                            // the dead ones behind it (zeroed at deposit)
                    const long long lastel = row_floats - 1;
                    v.x = base[e < lastel ? e : lastel];
                    v.y = base[e + 1 < lastel ? e + 1 : lastel];
                    v.z = base[e + 2 < lastel ? e + 2 : lastel];
                    v.w = base[e + 3 < lastel ? e + 3 : lastel];
                    nxt[j] = v;
                }
            } else {
                // generic frame stride: chunk c covers bins 4*(c % q4r) .. of frame c / q4r, q4r chunks per row
                const int q4r = (n_freqs + 3) >> 2;
                const int fi = c / q4r, b0 = 4 * (c - fi * q4r);
                long long frame = f0 + fi;
                frame = frame < n_frames ? frame : n_frames - 1;
                const float* src = base + frame * stride_t;
                fbs_f4 v;
                v.x = src[b0 < n_freqs ? b0 : n_freqs - 1];
                v.y = src[b0 + 1 < n_freqs ? b0 + 1 : n_freqs - 1];
                v.z = src[b0 + 2 < n_freqs ? b0 + 2 : n_freqs - 1];
                v.w = src[b0 + 3 < n_freqs ? b0 + 3 : n_freqs - 1];
                nxt[j] = v;
            }
        }
    };
    auto deposit = [&](int tile) {
        const int row = tile / tiles_per_row;
        const long long f0 = (long long)(tile - row * tiles_per_row) * FBS_TILE;
        const int live = (int)((n_frames - f0) < FBS_TILE ? (n_frames - f0) : FBS_TILE);
#pragma unroll
        for (int j = 0; j < FBS_CHUNKS; ++j) {
            const int c = tid + FBS_WAVES * 64 * j;
            int fi, bin;
            if (packed) {
                fi = (4 * c) / n_freqs;
                bin = 4 * c - fi * n_freqs;
            } else {
                const int q4r = (n_freqs + 3) >> 2;
                fi = c / q4r;
                bin = 4 * (c - fi * q4r);
            }
            const float vals[4] = {nxt[j].x, nxt[j].y, nxt[j].z, nxt[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (packed && bin >= n_freqs) { bin -= n_freqs; ++fi; }
                if (fi < FBS_TILE && bin < n_freqs) ptile[fi * prow_stride + bin] = fi < live ? vals[i] : 0.0f;
                ++bin;
            }
        }
    };
    // the contraction's 8-tap trips may read up to 7 bins past the row: zero them once
    for (int i = tid; i < FBS_TILE * 8; i += FBS_WAVES * 64) {
        const int col = n_freqs + (i & 7);
        if (col < prow_stride) ptile[(i >> 3) * prow_stride + col] = 0.0f;
    }
    if (begin < end) request(begin);
    for (int tile = begin; tile < end; ++tile) {
        const int row = tile / tiles_per_row;
        const long long f0 = (long long)(tile - row * tiles_per_row) * FBS_TILE;
        deposit(tile);                                         // rows of this tile (requested during the previous one)
        __syncthreads();
        if (tile + 1 < end) request(tile + 1);            // in flight during the contraction
        sparse_phase_b(dg, prow, wlds, otile + fr * ostr);
        __syncthreads();
        sparse_phase_c<FBS_WAVES * 64>(otile, ostr, tid, FBS_TILE, m, g, row, f0);
        // the barrier after the next deposit orders these otile reads before the next contraction's writes, and the
        // barrier above ordered this contraction's row reads before the next deposit
    }
}

template <int NC, int E>
static int sparse_groups() { return MelCfg<NC, E, SP_TILE>::WAVES * 4; }

static int sparse_groups_for(int n_fft) {
    switch (n_fft) {
        case 32: return sparse_groups<16, 16>();
        case 64: return sparse_groups<32, 16>();
        case 128: return sparse_groups<64, 16>();
        case 256: return sparse_groups<128, 16>();
        case 512: return sparse_groups<256, 16>();
        case 1024: return sparse_groups<512, 16>();
        case 2048: return 64;                                                // one band per lane and slot (melspec_stream.hpp)
        case 400: return LM_MARK + 8;                                        // eight lanes per frame (stft_n400.hip, mel_lanes.hpp)
        case 4096: return LM_MARK + 4096;                                    // == N4M_MARK (stft_n4096_s3.hpp)
        default: return 0;
    }
}

template <int NC, int E>
static int launch_sparse(const FrameGeom& g, const Tables& tb, const SparseArgs& m, float power, hipStream_t stream) {
    using C = MelCfg<NC, E, SP_TILE>;
    const int ostr = sparse_ostr(m.n_mels, m.out_vec4);
    const size_t lds_bytes = (size_t)C::NBUF * C::F::PADDED * sizeof(cf) + (size_t)m.wtot * 4 +
                             (size_t)((SP_TILE * ostr + 3) & ~3) * 4 + (size_t)C::WAVES * 4 * m.dstride * 4;
    if (lds_bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    const long long tiles = g.rows * ((g.n_frames + SP_TILE - 1) / SP_TILE);
    if (tiles >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    int per_cu = (int)(160 * 1024 / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    long long max_blocks = (long long)device_cu_count() * per_cu;
    long long blocks = tiles < max_blocks ? tiles : max_blocks;
    if (blocks < 1) blocks = 1;
    const bool pow2 = (power == 2.0f);
    auto kern = pow2 ? melspec_sparse_kernel<NC, E, true> : melspec_sparse_kernel<NC, E, false>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::WAVES * 64), lds_bytes, stream, g, tb, m);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// fft_length 2048: the barrier-free streaming kernel (melspec_stream.hpp), one persistent workgroup per CU.
// info_host: {weight floats, slots, 64, total steps, steps of slot 0..3} from tac_melbank_pack.
template <int NC, int E, int FMT>
static int launch_stream(FrameGeom g, const Tables& tb, const SparseArgs& sm, const int32_t* info_host, float power,
                         hipStream_t stream, const void* samples, const float* lut) {
    const long long total = g.rows * g.n_frames;
    if (total >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;                  // 32-bit global frame numbers in-kernel
    if (g.length < 2 * NC) return TAC_E_UNSUPPORTED;                       // (its clamped sample requests need a whole frame)
    const size_t lds_bytes = stream_lds_bytes<NC, E>(sm.wtot);
    if (info_host[1] < 1 || info_host[1] > ST_MAX_SLOTS) return TAC_E_UNSUPPORTED;
    if (FMT != FMT_F32) {                                                  // sample pairs fetched as one access of the format
        const uintptr_t pair = FMT == FMT_I16 ? 4 : (FMT == FMT_MULAW_U8 ? 2 : 8);
        g.vec2_ok = ((g.hop & 1) == 0) && ((g.center_pad & 1) == 0) && ((g.row_stride & 1) == 0) &&
                    ((reinterpret_cast<uintptr_t>(samples) & (pair - 1)) == 0);
    }
    StreamArgs m{sm.wpack, sm.desc, info_host[1], {info_host[4], info_host[5], info_host[6], info_host[7]}, sm.wtot,
                 sm.n_mels, sm.db, sm.amin, sm.log10_ref, sm.out, total, samples, lut, 0, nullptr, (info_host[2] & ST_REV_MARK) ? 1 : 0};
    long long blocks = (total + 2 * ST_WAVES - 1) / (2 * ST_WAVES);
    if (blocks > device_cu_count()) blocks = device_cu_count();
    if (blocks < 1) blocks = 1;
    const bool pow2 = (power == 2.0f);
    // the band predicate of the row stores is compiled out for whole slots of 64 bands (float32 input only: the coded
    // formats keep one instantiation per power)
    const bool fullm = FMT == FMT_F32 && (sm.n_mels % 64) == 0;
    // two slots of exactly (4, 16) steps — what tac_melbank_pack produces for 128-band mel banks — take the kernel whose
    // contraction is unrolled for that shape (its first reads ride along with the other frame's FFT stage)
    const bool fast2 = fullm && info_host[1] == 2 && info_host[4] == ST_FAST_STEPS0 &&
                       (info_host[5] == ST_FAST_STEPS1 || info_host[5] == ST_FAST_STEPS1_SHORT);
    const bool fshort = info_host[5] == ST_FAST_STEPS1_SHORT;
    // Round 3: the three-waves-per-SIMD form (melspec_stream3.hpp) is the route; TAC_STREAM2=1 selects round 2's two-frame
    // rotation (kept for A/B runs and for tools/stream_timing.py)
    static const bool two_waves = getenv("TAC_STREAM2") != nullptr;
    // 12 waves per CU (a 15-wave / 128-register form measured 11 % slower: tools/ablation/README.md)
    constexpr bool coded = FMT != FMT_F32;
    constexpr int waves3 = S3_WAVES;
    // Round 6: the transform on the matrix pipe (melspec_mfma.hpp; float32 samples).  tac_set_fft_pipe() / TAC_FFT_PIPE=valu|mfma select the form.
    if constexpr (FMT == FMT_F32 && NC == 1024) {
        const bool use_mfma = fft_pipe_mfma();
        constexpr int WM = 12;
        const size_t ldsm = mfma_lds_bytes<NC, E>(sm.wtot, WM);
        if (use_mfma && !two_waves && ldsm <= 160 * 1024) {
            void (*km)(FrameGeom, Tables, StreamArgs);
            if (fast2 && fshort) km = pow2 ? melspec_mfma_kernel<true, ST_FAST_STEPS1_SHORT, WM> : melspec_mfma_kernel<false, ST_FAST_STEPS1_SHORT, WM>;
            else if (fast2) km = pow2 ? melspec_mfma_kernel<true, ST_FAST_STEPS1, WM> : melspec_mfma_kernel<false, ST_FAST_STEPS1, WM>;
            else km = pow2 ? melspec_mfma_kernel<true, 0, WM> : melspec_mfma_kernel<false, 0, WM>;
            long long bm = (total + WM - 1) / WM;
            if (bm > device_cu_count()) bm = device_cu_count();
            m.chunk = (total + bm - 1) / bm;
            m.probe = (g_clock_probe && g_clock_probe_pairs >= bm) ? g_clock_probe : nullptr;
            set_last_route("melspec_mfma_kernel<%s, %d, %d>", pow2 ? "true" : "false", fast2 ? (fshort ? ST_FAST_STEPS1_SHORT : ST_FAST_STEPS1) : 0, WM);
            TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(km), 160 * 1024));
            hipLaunchKernelGGL(km, dim3((unsigned)bm), dim3(WM * 64), ldsm, stream, g, tb, m);
            TAC_HIP(hipGetLastError());
            return TAC_OK;
        }
    }
    const size_t lds3 = stream3_lds_bytes<NC, E>(sm.wtot, waves3, coded) + ((FMT == FMT_F32 && fast2) ? S3_TW2L_BYTES : 0);   // (FAST1 kernels: + their pass-2 twiddle table)
    if (!two_waves && lds3 <= 160 * 1024) {
        void (*k3)(FrameGeom, Tables, StreamArgs);
        if constexpr (FMT == FMT_F32) {
            {
                constexpr int W = S3_WAVES;
                if (fast2 && fshort) k3 = pow2 ? melspec_stream3_kernel<NC, E, true, FMT, ST_FAST_STEPS1_SHORT, W> : melspec_stream3_kernel<NC, E, false, FMT, ST_FAST_STEPS1_SHORT, W>;
                else if (fast2) k3 = pow2 ? melspec_stream3_kernel<NC, E, true, FMT, ST_FAST_STEPS1, W> : melspec_stream3_kernel<NC, E, false, FMT, ST_FAST_STEPS1, W>;
                else k3 = pow2 ? melspec_stream3_kernel<NC, E, true, FMT, 0, W> : melspec_stream3_kernel<NC, E, false, FMT, 0, W>;
            }
        } else {
            k3 = pow2 ? melspec_stream3_kernel<NC, E, true, FMT, 0, S3_WAVES> : melspec_stream3_kernel<NC, E, false, FMT, 0, S3_WAVES>;
        }
        long long b3 = (total + waves3 - 1) / waves3;
        if (b3 > device_cu_count()) b3 = device_cu_count();
        m.chunk = (total + b3 - 1) / b3;
        m.probe = (g_clock_probe && g_clock_probe_pairs >= b3) ? g_clock_probe : nullptr;
        {
            const int fast1 = (FMT == FMT_F32 && fast2) ? (fshort ? ST_FAST_STEPS1_SHORT : ST_FAST_STEPS1) : 0;
            set_last_route("melspec_stream3_kernel<%d, %d, %s, %d, %d, %d>", NC, E, pow2 ? "true" : "false", FMT, fast1, waves3);
        }
        TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(k3), 160 * 1024));
        hipLaunchKernelGGL(k3, dim3((unsigned)b3), dim3(waves3 * 64), lds3, stream, g, tb, m);
        TAC_HIP(hipGetLastError());
        return TAC_OK;
    }
    void (*kern)(FrameGeom, Tables, StreamArgs);
    if constexpr (FMT == FMT_F32) {
        if (fast2 && fshort)
            kern = pow2 ? melspec_stream_kernel<NC, E, true, true, FMT, ST_FAST_STEPS1_SHORT> : melspec_stream_kernel<NC, E, false, true, FMT, ST_FAST_STEPS1_SHORT>;
        else if (fast2)
            kern = pow2 ? melspec_stream_kernel<NC, E, true, true, FMT, ST_FAST_STEPS1> : melspec_stream_kernel<NC, E, false, true, FMT, ST_FAST_STEPS1>;
        else
            kern = pow2 ? (fullm ? melspec_stream_kernel<NC, E, true, true, FMT, 0> : melspec_stream_kernel<NC, E, true, false, FMT, 0>)
                        : (fullm ? melspec_stream_kernel<NC, E, false, true, FMT, 0> : melspec_stream_kernel<NC, E, false, false, FMT, 0>);
    } else {
        kern = pow2 ? melspec_stream_kernel<NC, E, true, false, FMT, 0> : melspec_stream_kernel<NC, E, false, false, FMT, 0>;
    }
    if (lds_bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(ST_WAVES * 64), lds_bytes, stream, g, tb, m);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// ---------------------------------------------------------------- standalone band-sparse filterbank, wave-autonomous form
// functional.apply_filterbank on a frame-major spectrogram whose bank fits the lane layout of mel_lanes.hpp with one
// frame per wave (LANES = 64: lane l owns bands l and 64 + l): every wave draws frames from a workgroup counter, parks
// the frame's row in its own LDS buffer (16-byte chunks, the next frame's requested before this one is contracted),
// contracts it and stores the band row — no workgroup barrier, no tile.  (fb_sparse_kernel above, round 1's 16-frame
// tiles between barriers, stays the route for banks outside this layout and for other frame strides.)
// Rows up to 1280 bins (fft_length <= 2560): SIXTEEN 128-register waves per CU with 8 contraction steps in flight (round 3:
// 0.092 -> 0.080 ms on cfg-2's 1025 x 128, 4.0 -> 4.6 TB/s; twelve waves 0.081); wider rows (fft_length 4096): eight waves, 16 steps
// in flight — sixteen row buffers of 8 KB do not fit next to 41 KB of weights.  The packed layout depends on the steps in flight
// (lm_group), so the pack and the launch use the same rule.
constexpr int FBL_CHUNKS = 5, FBL_CHUNKS_WIDE = 9;   // 16-byte chunks per lane and frame: up to 1280 / 2304 bins (fft_length 2048 / 4096)
__host__ __device__ inline bool fbl_is_wide(int n_freqs) { return (n_freqs + 3) / 4 > FBL_CHUNKS * 64; }
__host__ __device__ inline int fbl_waves(int n_freqs) { return fbl_is_wide(n_freqs) ? 8 : 16; }
__host__ __device__ inline int fbl_fly(int n_freqs) { return fbl_is_wide(n_freqs) ? 16 : 8; }
__host__ __device__ inline int fbl_pitch(int n_freqs) { return (n_freqs + 3 + 3) & ~3; }
inline size_t fbl_base_lds(int n_freqs) { return (size_t)fbl_waves(n_freqs) * (fbl_pitch(n_freqs) + LM_MAX_MELS + 4) * sizeof(float) + 16; }

template <int S, int CHUNKS, int FBL_WAVES, int FBL_FLY>
__global__ void __launch_bounds__(FBL_WAVES * 64, FBL_WAVES / 4)
fb_lanes_kernel(const float* __restrict__ spec, long long rows, int n_freqs, long long n_frames, long long stride_r,
                long long stride_t, LaneMel mel) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pitch = fbl_pitch(n_freqs);
    float* const srow = reinterpret_cast<float*>(smem_raw) + (size_t)w * (pitch + LM_MAX_MELS + 4);
    float* const mbuf = srow + pitch;                                        // the band row, staged at its 16-byte phase
    unsigned* const next_frame = reinterpret_cast<unsigned*>(reinterpret_cast<float*>(smem_raw) +
                                                             (size_t)FBL_WAVES * (pitch + LM_MAX_MELS + 4));
    int* const mlo = reinterpret_cast<int*>(next_frame + 4);
    float* const mwl = reinterpret_cast<float*>(mlo + lm_desc_ints(64));
    lane_mel_load_tables<S, 64, FBL_FLY>(mlo, mwl, mel, threadIdx.x, FBL_WAVES * 64);

    const long long total = rows * n_frames;
    const long long chunk = (total + gridDim.x - 1) / gridDim.x;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < total ? begin + chunk : total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    if (threadIdx.x == 0) *next_frame = FBL_WAVES;
    __syncthreads();
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    // a frame's row as 16-byte chunks (global loads only need dword alignment); chunks past the row are clamped to its
    // last full one (the slack behind the bins is zeroed by the contraction)
    const int nch = (n_freqs + 3) >> 2, lastc = (n_freqs >> 2) - 1;         // chunk `nch - 1` may straddle the row's end
    f4 nxt[CHUNKS];
    float tail[3];
    auto request = [&](int i) {
        const long long gf = begin + i;
        const long long r = gf / n_frames;
        const float* src = spec + r * stride_r + (gf - r * n_frames) * stride_t;
#pragma unroll
        for (int u = 0; u < CHUNKS; ++u) {
            const int c = lane + 64 * u;
            nxt[u] = *reinterpret_cast<const f4*>(src + 4 * (c < lastc ? c : lastc));
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) tail[u] = src[n_freqs - 1 - u];          // the (n_freqs mod 4) bins behind the last full chunk
    };
    auto deposit = [&]() {
#pragma unroll
        for (int u = 0; u < CHUNKS; ++u) {
            const int c = lane + 64 * u;
            if (c <= lastc) *reinterpret_cast<f4*>(srow + 4 * c) = nxt[u];
        }
        if (lane < 3 && n_freqs - 1 - lane > 4 * lastc + 3) srow[n_freqs - 1 - lane] = tail[lane == 0 ? 0 : (lane == 1 ? 1 : 2)];
    };
    int i = w;
    if (i < nloc) request(i);
    while (i < nloc) {
        const int nx = grab();
        wave_lds_fence();
        deposit();
        wave_lds_fence();
        if (nx < nloc) request(nx);                                           // in flight during the contraction
        const long long g0 = (begin + i) * (long long)mel.n_mels;
        const int am = (int)(g0 & 3);
        lane_mel_contract<S, 64, FBL_FLY>(srow, n_freqs, mlo, mwl, lane, mel, mbuf + am);
        wave_lds_fence();
        lane_mel_store<1>(mbuf + am, am, mel.n_mels, mel.out + g0, lane);
        i = nx;
    }
}

template <int S, int CHUNKS>
static int launch_fb_lanes(const float* spec, long long rows, int n_freqs, long long n_frames, long long stride_r,
                           long long stride_t, const LaneMel& mel, hipStream_t stream) {
    constexpr int FBL_WAVES = CHUNKS == FBL_CHUNKS_WIDE ? 8 : 16, FBL_FLY = CHUNKS == FBL_CHUNKS_WIDE ? 16 : 8;
    const size_t bytes = fbl_base_lds(n_freqs) + lm_lds_bytes(64, mel.wtot);
    if (bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    const long long total = rows * n_frames;
    if (total >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    long long blocks = (total + FBL_WAVES - 1) / FBL_WAVES;
    if (blocks > device_cu_count()) blocks = device_cu_count();
    auto kern = fb_lanes_kernel<S, CHUNKS, FBL_WAVES, FBL_FLY>;
    if (bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(FBL_WAVES * 64), bytes, stream, spec, rows, n_freqs, n_frames, stride_r,
                       stride_t, mel);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// Lane layout of the streaming kernel: cell c = 64 s + l (slot s, lane l) holds band c — or, for banks whose band count is not a
// multiple of 64 and whose bands widen with their number (every mel bank), band n_mels - 1 - c (round 6, ST_REV_MARK in info[2]): the
// widest bands then share slot 0 instead of defining a mostly empty last slot (80 bands at 16 kHz: 24 steps instead of 32).  Every
// slot is one loop of steps[s] four-tap steps (the longest band of the slot, in whole trips of four steps); shorter bands are
// zero-padded, and a band whose padded run would leave the row buffer is shifted down (zeros in front) so that every
// lane reads inside its row.
static int pack_lanes(const std::vector<float>& h, int n_freqs, int n_mels, float* wpack, int wpack_cap, int32_t* desc,
                      int desc_cap, int32_t* info_host, hipStream_t stream, bool to_host = false) {
    const int nslot = (n_mels + 63) / 64;
    if (nslot > ST_MAX_SLOTS || nslot * 64 > desc_cap) return TAC_E_UNSUPPORTED;
    const int limit = StreamCfg<1024, 16>::PROW;                             // bins + zeroed slack of a row buffer
    std::vector<int> blo(n_mels, 0), blen(n_mels, 0);                        // per band: first bin (multiple of four), bins from there
    for (int m = 0; m < n_mels; ++m) {
        int l0 = n_freqs, h0 = 0;
        for (int f = 0; f < n_freqs; ++f)
            if (h[(size_t)f * n_mels + m] != 0.0f) { l0 = f < l0 ? f : l0; h0 = f + 1; }
        if (h0 > l0) {
            blo[m] = l0 & ~3;
            blen[m] = h0 - blo[m];
        }
    }
    const bool rev = (n_mels % 64) != 0 && blen[n_mels - 1] > blen[0];
    auto band_of = [&](int c) { return rev ? n_mels - 1 - c : c; };
    std::vector<int> lo(nslot * 64, 0), len(nslot * 64, 0);                  // per cell
    int steps[ST_MAX_SLOTS] = {0, 0, 0, 0};
    for (int c = 0; c < n_mels; ++c) {
        lo[c] = blo[band_of(c)];
        len[c] = blen[band_of(c)];
        steps[c / 64] = std::max(steps[c / 64], (len[c] + 3) / 4);
    }
    const bool fast_shape = nslot == 2 && (n_mels % 64) == 0 && steps[0] <= ST_FAST_STEPS0 && steps[1] <= ST_FAST_STEPS1;
    if (fast_shape) {                                                        // the shapes the FAST2 kernels are unrolled for
        steps[0] = ST_FAST_STEPS0;
        steps[1] = steps[1] <= ST_FAST_STEPS1_SHORT ? ST_FAST_STEPS1_SHORT : ST_FAST_STEPS1;
    }
    int total_steps = 0;
    for (int s = 0; s < nslot; ++s) {
        if (!fast_shape) steps[s] = std::max(4, (steps[s] + 3) & ~3);        // whole trips; an empty slot still runs one
        if (4 * steps[s] > limit) return TAC_E_UNSUPPORTED;
        total_steps += steps[s];
    }
    const long long wtot = 256LL * total_steps;
    // (the twelve-wave kernel's LDS decides: until round 6 this asked the two-waves-per-SIMD kernel's formula, which refused every
    // table above ~20 steps — 80- and 96-band banks at 2048 took the two-launch chain for no reason)
    if (wtot > wpack_cap || stream3_lds_bytes<1024, 16>((int)wtot, S3_WAVES, true) > 160 * 1024) return TAC_E_UNSUPPORTED;
    std::vector<float> wp((size_t)wtot, 0.0f);
    // bank-aware placement of the runs' first bins (lane_placement.hpp)
    const std::vector<int> start = place_band_starts(nslot, steps, lo, len);
    int base = 0;
    for (int s = 0; s < nslot; ++s) {
        for (int l = 0; l < 64; ++l) {
            const int c = s * 64 + l, m = c < n_mels ? band_of(c) : 0;
            int first = start[c];
            if (first + 4 * steps[s] > limit) first = (limit - 4 * steps[s]) & ~3;     // keep the padded run inside the row
            for (int j = 0; j < steps[s]; ++j)
                for (int u = 0; u < 4; ++u) {
                    const int bin = first + 4 * j + u;
                    const bool live = c < n_mels && bin >= lo[c] && bin < lo[c] + len[c] && bin < n_freqs;
                    wp[((size_t)(base + j) * 64 + l) * 4 + u] = live ? h[(size_t)bin * n_mels + m] : 0.0f;
                }
            lo[c] = first;
        }
        base += steps[s];
    }
    if (to_host) {                                                           // (tac_melbank_pack_host: host buffers, no device)
        std::copy(wp.begin(), wp.end(), wpack);
        std::copy(lo.begin(), lo.end(), desc);
    } else {
        TAC_HIP(hipMemcpyAsync(wpack, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        TAC_HIP(hipMemcpyAsync(desc, lo.data(), lo.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        TAC_HIP(hipStreamSynchronize(stream));
    }
    info_host[0] = (int32_t)wtot;
    info_host[1] = nslot;
    info_host[2] = 64 + (rev ? ST_REV_MARK : 0);
    info_host[3] = total_steps;
    for (int s = 0; s < ST_MAX_SLOTS; ++s) info_host[4 + s] = steps[s];
    return TAC_OK;
}

}  // namespace tac

extern "C" {

int tac_melbank_plan_pieces_host(const float* fb_host, int32_t n_freqs, int32_t n_mels, int32_t* seg_steps, int32_t* first,
                                 int32_t* band, int32_t* index, float* weights, int32_t weights_cap) {
    using namespace tac;
    if (!fb_host || !seg_steps || !first || !band || !index || !weights || n_freqs <= 0 || n_mels <= 0) return TAC_E_INVALID;
    PiecePlan p;
    if (!plan_pieces(fb_host, n_freqs, n_mels, StreamCfg<1024, 16>::PROW, &p)) return TAC_E_UNSUPPORTED;
    if ((long long)p.w.size() > weights_cap) return TAC_E_INVALID;
    for (int s = 0; s < MP_SEGS; ++s) seg_steps[s] = p.L[s];
    for (int e = 0; e < MP_SEGS * 64; ++e) {
        first[e] = p.first[e];
        band[e] = p.band[e];
        index[e] = p.index[e];
    }
    std::copy(p.w.begin(), p.w.end(), weights);
    return TAC_OK;
}

int tac_melbank_pack_host(const float* fb_host, int32_t n_freqs, int32_t n_mels, int32_t n_fft, float* wpack_host, int32_t wpack_cap,
                          int32_t* desc_host, int32_t desc_cap, int32_t* info_host) {
    using namespace tac;
    if (!fb_host || !wpack_host || !desc_host || !info_host || n_freqs <= 0 || n_mels <= 0) return TAC_E_INVALID;
    const bool lanes = n_fft == 256 || n_fft == 400 || n_fft == 512 || n_fft == 1024;
    if ((n_fft != 2048 && n_fft != 4096 && !lanes) || n_freqs != n_fft / 2 + 1) return TAC_E_UNSUPPORTED;
    const std::vector<float> h(fb_host, fb_host + (size_t)n_freqs * n_mels);
    if (n_fft == 400) return pack_n400(h, n_freqs, n_mels, wpack_host, wpack_cap, desc_host, desc_cap, info_host, nullptr, true);
    if (lanes) return pack_small(n_fft, h, n_freqs, n_mels, wpack_host, wpack_cap, desc_host, desc_cap, info_host, nullptr, true);
    return n_fft == 2048 ? pack_lanes(h, n_freqs, n_mels, wpack_host, wpack_cap, desc_host, desc_cap, info_host, nullptr, true)
                         : pack_n4096_mel(h, n_freqs, n_mels, wpack_host, wpack_cap, desc_host, desc_cap, info_host, nullptr, true);
}

int tac_melbank_pack(const float* fb, int32_t n_freqs, int32_t n_mels, int32_t n_fft, float* wpack,
                     int32_t wpack_cap, int32_t* desc, int32_t desc_cap, int32_t* info_host, void* stream) {
    using namespace tac;
    if (!fb || !wpack || !desc || !info_host || n_freqs <= 0 || n_mels <= 0) return TAC_E_INVALID;
    if (n_fft < 0) return TAC_E_UNSUPPORTED;          // (round 4's piece layout, TAC_PACK_PIECES_2048: its kernel form is not shipped)
    // n_fft == 0: pack for the standalone filterbank kernel (32 lane groups, any number of bins)
    const int groups = n_fft == 0 ? FBS_WAVES * 4 : sparse_groups_for(n_fft);
    if (groups == 0 || (n_fft != 0 && n_freqs != n_fft / 2 + 1)) return TAC_E_UNSUPPORTED;
    std::vector<float> h((size_t)n_freqs * n_mels);
    TAC_HIP(hipMemcpyAsync(h.data(), fb, h.size() * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    TAC_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (n_fft == 2048)
        return pack_lanes(h, n_freqs, n_mels, wpack, wpack_cap, desc, desc_cap, info_host, (hipStream_t)stream);
    if (n_fft == 4096)                                                      // the one-launch chain of stft_n4096_s3.hpp
        return pack_n4096_mel(h, n_freqs, n_mels, wpack, wpack_cap, desc, desc_cap, info_host, (hipStream_t)stream);
    if (n_fft == 0 && n_freqs >= 8 && (n_freqs + 3) / 4 <= FBL_CHUNKS_WIDE * 64) {   // standalone: one frame per wave
        const int rc = pack_lane_mel(h, n_freqs, n_mels, 64, fbl_pitch(n_freqs), 2, fbl_fly(n_freqs), LM_MAX_STEPS_WAVE, fbl_base_lds(n_freqs),
                                     wpack, wpack_cap, desc, desc_cap, info_host, (hipStream_t)stream);
        if (rc != TAC_E_UNSUPPORTED) return rc;                             // else: the tile kernel's layout
    }
    if (n_fft == 400) return pack_n400(h, n_freqs, n_mels, wpack, wpack_cap, desc, desc_cap, info_host, (hipStream_t)stream);
    if (n_fft == 256 || n_fft == 512 || n_fft == 1024) {
        const int rc = pack_small(n_fft, h, n_freqs, n_mels, wpack, wpack_cap, desc, desc_cap, info_host, (hipStream_t)stream);
        if (rc != TAC_E_UNSUPPORTED) return rc;                             // else: the three-phase kernel's layout
    }
    struct Band { int m, lo, len; };
    std::vector<Band> bands(n_mels);
    long long total = 0;
    for (int m = 0; m < n_mels; ++m) {
        int lo = n_freqs, hi = 0;
        for (int f = 0; f < n_freqs; ++f)
            if (h[(size_t)f * n_mels + m] != 0.0f) { lo = f < lo ? f : lo; hi = f + 1; }
        lo &= ~1;                                            // even first bin: the kernel reads the power row 8 bytes at a time
        bands[m] = {m, hi > lo ? lo : 0, hi > lo ? hi - lo : 0};
        total += (bands[m].len + 7) & ~7;
    }
    if (total > SP_MAX_W || total > wpack_cap) return TAC_E_UNSUPPORTED;      // not band-sparse enough for LDS
    // longest-processing-time dealing: sorted descending, each band to the currently lightest group
    std::stable_sort(bands.begin(), bands.end(), [](const Band& a, const Band& b) { return a.len > b.len; });
    std::vector<std::vector<Band>> per(groups);
    std::vector<int> load(groups, 0);
    for (const Band& b : bands) {
        int best = 0;
        for (int gI = 1; gI < groups; ++gI)
            if (load[gI] < load[best] || (load[gI] == load[best] && per[gI].size() < per[best].size())) best = gI;
        per[best].push_back(b);
        load[best] += ((b.len + 7) & ~7) + 4;      // +4: per-band loop overhead in "element" units
    }
    size_t maxnb = 0;
    for (auto& v : per) maxnb = std::max(maxnb, v.size());
    const int dstride = 4 + 4 * (int)maxnb;
    if ((long long)groups * dstride > desc_cap) return TAC_E_UNSUPPORTED;
    std::vector<float> wp((size_t)total, 0.0f);
    std::vector<int32_t> dd((size_t)groups * dstride, 0);
    int woff = 0, maxload = 0;
    for (int gI = 0; gI < groups; ++gI) {
        dd[(size_t)gI * dstride] = (int)per[gI].size();
        int gl = 0;
        for (size_t bI = 0; bI < per[gI].size(); ++bI) {
            const Band& b = per[gI][bI];
            const int n8 = (b.len + 7) / 8;
            int32_t* e = &dd[(size_t)gI * dstride + 4 + 4 * bI];
            e[0] = b.m; e[1] = b.lo; e[2] = n8; e[3] = woff;
            for (int j = 0; j < b.len; ++j) wp[woff + j] = h[(size_t)(b.lo + j) * n_mels + b.m];
            woff += 8 * n8;
            gl += 8 * n8;
        }
        maxload = std::max(maxload, gl);
    }
    TAC_HIP(hipMemcpyAsync(wpack, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    TAC_HIP(hipMemcpyAsync(desc, dd.data(), dd.size() * sizeof(int32_t), hipMemcpyHostToDevice, (hipStream_t)stream));
    TAC_HIP(hipStreamSynchronize((hipStream_t)stream));
    info_host[0] = (int32_t)total;
    info_host[1] = dstride;
    info_host[2] = groups;
    info_host[3] = maxload;
    return TAC_OK;
}

int tac_melspec_sparse_f32(const float* wave, const float* window, const tac_stft_desc* d, float power,
                           const float* wpack, const int32_t* desc, const int32_t* info_host, int32_t n_mels,
                           int db, float db_ref, float db_amin, float* out, void* stream) {
    using namespace tac;
    if (!wpack || !desc || !info_host || !out || !d || n_mels <= 0) return TAC_E_INVALID;
    if (!d->onesided || d->n_fft > 4096) return TAC_E_UNSUPPORTED;
    if (power != 2.0f && power != 1.0f) return TAC_E_UNSUPPORTED;
    if (d->n_fft == 4096) {                                                         // stft_n4096.hip: the twelve-wave form + contraction
        FrameGeom g4;
        int64_t T4 = 0;
        const int rc4 = make_geometry(wave, window, d, &g4, &T4);
        if (rc4 != TAC_OK) return rc4;
        return launch_n4096_mel(g4, power, wpack, desc, info_host, n_mels, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f, out,
                                (hipStream_t)stream);
    }
    const bool lanes_pack = info_host[2] >= LM_MARK;                               // mel_lanes.hpp layout
    if (!lanes_pack && (info_host[2] & ~(d->n_fft == 2048 ? ST_REV_MARK : 0)) != sparse_groups_for(d->n_fft)) return TAC_E_INVALID;   // pack built for another geometry
    FrameGeom g;
    int64_t T = 0;
    int rc = make_geometry(wave, window, d, &g, &T);
    if (rc != TAC_OK) return rc;
    if (lanes_pack && (d->n_fft == 256 || d->n_fft == 512 || d->n_fft == 1024)) {
        Tables tbs;
        rc = get_tables(d->n_fft, &tbs);
        if (rc != TAC_OK) return rc;
        return launch_small_mel_entry(d->n_fft, g, tbs, power, wpack, desc, info_host, n_mels, db ? 1 : 0, db_amin,
                                      db ? log10f(db_ref) : 0.0f, out, (hipStream_t)stream);
    }
    if (lanes_pack && d->n_fft != 400) return TAC_E_INVALID;
    if (d->n_fft == 400)
        return launch_n400_mel(g, power, wpack, desc, info_host, n_mels, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f, out,
                               (hipStream_t)stream);
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    const int out_vec4 = ((n_mels & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    SparseArgs m{wpack, desc, info_host[0], info_host[1], n_mels, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f, out, out_vec4};
    hipStream_t s = (hipStream_t)stream;
    switch (d->n_fft) {
        case 32: return launch_sparse<16, 16>(g, tb, m, power, s);
        case 64: return launch_sparse<32, 16>(g, tb, m, power, s);
        case 128: return launch_sparse<64, 16>(g, tb, m, power, s);
        case 256: return launch_sparse<128, 16>(g, tb, m, power, s);
        case 512: return launch_sparse<256, 16>(g, tb, m, power, s);
        case 1024: return launch_sparse<512, 16>(g, tb, m, power, s);
        case 2048: return launch_stream<1024, 16, FMT_F32>(g, tb, m, info_host, power, s, wave, nullptr);
        default: return TAC_E_UNSUPPORTED;
    }
}

int tac_melspec_sparse_coded_f32(const void* samples, int32_t sample_format, const float* decode_lut, const float* window,
                                 const tac_stft_desc* d, float power, const float* wpack, const int32_t* desc,
                                 const int32_t* info_host, int32_t n_mels, int db, float db_ref, float db_amin, float* out,
                                 void* stream) {
    using namespace tac;
    if (!samples || !wpack || !desc || !info_host || !out || !d || n_mels <= 0) return TAC_E_INVALID;
    if (sample_format < TAC_SAMPLES_F32 || sample_format > TAC_SAMPLES_MULAW_I64) return TAC_E_INVALID;
    if (sample_format >= TAC_SAMPLES_MULAW_U8 && !decode_lut) return TAC_E_INVALID;
    const bool small = d->n_fft == 256 || d->n_fft == 512 || d->n_fft == 1024 || d->n_fft == 400;   // lane-layout packs
    if (!d->onesided || (d->n_fft != 2048 && !small)) return TAC_E_UNSUPPORTED;
    if (power != 2.0f && power != 1.0f) return TAC_E_UNSUPPORTED;
    if (small ? info_host[2] < LM_MARK : (info_host[2] & ~ST_REV_MARK) != sparse_groups_for(d->n_fft)) return small ? TAC_E_UNSUPPORTED : TAC_E_INVALID;
    FrameGeom g;
    int64_t T = 0;
    int rc = make_geometry(static_cast<const float*>(samples), window, d, &g, &T);
    if (rc != TAC_OK) return rc;
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    if (small) {
        if (sample_format == TAC_SAMPLES_F32) return TAC_E_UNSUPPORTED;   // (float32 callers use tac_melspec_sparse_f32)
        if (d->n_fft == 400)
            return launch_n400_mel(g, power, wpack, desc, info_host, n_mels, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f, out,
                                   (hipStream_t)stream, sample_format, samples, decode_lut);
        return launch_small_mel_entry(d->n_fft, g, tb, power, wpack, desc, info_host, n_mels, db ? 1 : 0, db_amin,
                                      db ? log10f(db_ref) : 0.0f, out, (hipStream_t)stream, sample_format, samples, decode_lut);
    }
    SparseArgs m{wpack, desc, info_host[0], info_host[1], n_mels, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f, out, 0};
    hipStream_t s = (hipStream_t)stream;
    switch (sample_format) {
        case TAC_SAMPLES_F32: return launch_stream<1024, 16, FMT_F32>(g, tb, m, info_host, power, s, samples, nullptr);
        case TAC_SAMPLES_I16: return launch_stream<1024, 16, FMT_I16>(g, tb, m, info_host, power, s, samples, nullptr);
        case TAC_SAMPLES_MULAW_U8: return launch_stream<1024, 16, FMT_MULAW_U8>(g, tb, m, info_host, power, s, samples, decode_lut);
        default: return launch_stream<1024, 16, FMT_MULAW_I64>(g, tb, m, info_host, power, s, samples, decode_lut);
    }
}

int tac_apply_filterbank_sparse_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames,
                                    int64_t stride_r, int64_t stride_t, const float* wpack, const int32_t* desc,
                                    const int32_t* info_host, int32_t n_mels, float* out, void* stream) {
    return tac_apply_filterbank_sparse_db_f32(spec, rows, n_freqs, n_frames, stride_r, stride_t, wpack, desc, info_host, n_mels, 0,
                                              1.0f, 1e-7f, out, stream);
}

int tac_apply_filterbank_sparse_db_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames,
                                       int64_t stride_r, int64_t stride_t, const float* wpack, const int32_t* desc,
                                       const int32_t* info_host, int32_t n_mels, int db, float db_ref, float db_amin,
                                       float* out, void* stream) {
    using namespace tac;
    if (rows == 0 || n_frames == 0) return TAC_OK;
    if (!spec || !wpack || !desc || !info_host || !out) return TAC_E_INVALID;
    if (rows < 0 || n_freqs <= 0 || n_frames < 0 || n_mels <= 0) return TAC_E_INVALID;
    if (db && !(db_ref > 0.0f)) return TAC_E_INVALID;
    const float log10_ref = db ? log10f(db_ref) : 0.0f;
    if (info_host[2] == LM_MARK + 64) {                                            // lane layout: the wave-autonomous kernel
        if (!lane_mel_info_ok(info_host, 64, fbl_fly(n_freqs), LM_MAX_STEPS_WAVE)) return TAC_E_INVALID;
        const int chunks = (n_freqs + 3) / 4;
        if (n_mels < LM_MIN_MELS || n_mels > LM_MAX_MELS || chunks > FBL_CHUNKS_WIDE * 64) return TAC_E_UNSUPPORTED;
        const LaneMel lm{wpack, desc, info_host[1], info_host[0], n_mels, db ? 1 : 0, db_amin, log10_ref, out, info_host[5] ? 1 : 0};
        const long long sr = rows > 1 ? stride_r : 0;
        const bool wide = chunks > FBL_CHUNKS * 64;
        switch (info_host[4]) {
#define TAC_FBL_CASE(SS)                                                                                                    \
    case SS:                                                                                                               \
        return wide ? launch_fb_lanes<SS, FBL_CHUNKS_WIDE>(spec, rows, n_freqs, n_frames, sr, stride_t, lm, (hipStream_t)stream) \
                    : launch_fb_lanes<SS, FBL_CHUNKS>(spec, rows, n_freqs, n_frames, sr, stride_t, lm, (hipStream_t)stream);
            TAC_FBL_CASE(2) TAC_FBL_CASE(4) TAC_FBL_CASE(6) TAC_FBL_CASE(8) TAC_FBL_CASE(10) TAC_FBL_CASE(12) TAC_FBL_CASE(14)
            TAC_FBL_CASE(16) TAC_FBL_CASE(18) TAC_FBL_CASE(20) TAC_FBL_CASE(22) TAC_FBL_CASE(24) TAC_FBL_CASE(26) TAC_FBL_CASE(28)
            TAC_FBL_CASE(30) TAC_FBL_CASE(32) TAC_FBL_CASE(34) TAC_FBL_CASE(36)
#undef TAC_FBL_CASE
            default: return TAC_E_INVALID;
        }
    }
    if (info_host[2] != FBS_WAVES * 4) return TAC_E_INVALID;                       // pack built for another geometry
    if ((long long)FBS_TILE * ((n_freqs + 3) / 4) > (long long)FBS_CHUNKS * FBS_WAVES * 64) return TAC_E_UNSUPPORTED;
    int prow = n_freqs + 7;
    prow += (34 - (prow & 31)) & 31;                                               // == 2 (mod 32): conflict-free, 8-byte rows
    const int out_vec4 = ((n_mels & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    SparseArgs m{wpack, desc, info_host[0], info_host[1], n_mels, db ? 1 : 0, db_amin, log10_ref, out, out_vec4};
    const int ostr = sparse_ostr(n_mels, out_vec4);
    const size_t lds_bytes = (size_t)FBS_TILE * prow * 4 + (size_t)m.wtot * 4 + (size_t)((FBS_TILE * ostr + 3) & ~3) * 4 +
                             (size_t)FBS_WAVES * 4 * m.dstride * 4;
    if (lds_bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    const long long tiles = rows * ((n_frames + FBS_TILE - 1) / FBS_TILE);
    if (tiles >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    int per_cu = (int)(160 * 1024 / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    long long blocks = (long long)device_cu_count() * per_cu;
    if (blocks > tiles) blocks = tiles;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(fb_sparse_kernel), 160 * 1024));
    hipLaunchKernelGGL(fb_sparse_kernel, dim3((unsigned)blocks), dim3(FBS_WAVES * 64), lds_bytes, (hipStream_t)stream, spec,
                       (long long)rows, (int)n_freqs, (long long)n_frames, (long long)stride_r, (long long)stride_t, prow,
                       m);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

}  // extern "C"
