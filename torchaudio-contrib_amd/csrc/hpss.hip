// hpss.hip — harmonic / percussive separation of a magnitude spectrogram by median filtering (reference
// torchaudio_contrib/beta_hpss.py:35-127): the percussive-enhanced spectrogram is the running median
// along frequency, the harmonic-enhanced one the running median along time (reflect padding, `power` applied to both),
// from which soft ((h + eps) / (h + p + eps)) or hard (h > p) masks and the masked spectrograms follow.  The reference
// loops over columns / rows calling torch.median.
//
// Equal odd widths 9 ... 31 (the reference only runs with equal widths, default 31): hpss_tile8_kernel — a 256-thread
// workgroup owns a 64 x 64 tile of one spectrogram, staged ONCE in LDS with its reflect-padded halo (the fast memory axis is
// the LDS column axis, so the fill is coalesced whatever the layout: contiguous (F, T) or the frame-major strided views the
// STFT kernels return).  EIGHT consecutive windows along an axis share K - 7 of their K taps: sorted once in registers
// (Batcher network), merged with four more for each half of the run, each window's median selected from four ranks of that
// and its three own taps (median_run.hpp: 56 min / max operations per median at K = 31; a full 32-sort per output is 382).
// A NaN anywhere in a window makes its median NaN, as torch.median does (the min / max network alone would drop it).
// Any other odd widths <= 63 (33 ... 63 since round 5: the default of the reference is 31, but its argument is free):
// hpss_axis_a_kernel + hpss_axis_b_kernel, the same tiles with one halo axis per launch.
// hpss_tile_kernel (round 3: runs of four, 4 x 4 outputs per thread, dword accesses) and hpss_kernel (round 3: one thread per
// element) are kept as the A/B baselines (-DTAC_HPSS_RUN8=0, -DTAC_HPSS_TWO_PASS=0) and for non-unit fast strides.
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

#define HP_PROBE_BEGIN()
#define HP_PROBE_END()
// Tile of a workgroup, XCD-aware: workgroups go to the eight XCDs round-robin (block b -> XCD b % 8), and each XCD has
// its own L2.  With tile = block, spatial neighbours always sit on DIFFERENT XCDs and every halo is fetched from HBM again (rocprofv3:
// 873 MB read per launch for a 328 MB spectrogram); giving every XCD a contiguous eighth of the tiles — tile = (b % 8) * ceil(total / 8)
// + b / 8, the grid rounded up to a multiple of eight — keeps neighbours behind one L2.  Correct for any placement; only speed depends on it.
__device__ __forceinline__ long long hp_tile_of_block(long long total) {
    const long long per_xcd = (total + 7) / 8;
    return (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
}
static inline unsigned hp_grid_for(long long total) { return (unsigned)(((total + 7) / 8) * 8); }

// hpss_tile8_kernel (round 4): the same tile with
//  * runs of EIGHT windows per thread (median_run8: 56 instead of 97 min / max per median at K = 31), the NaN bookkeeping only
//    in tiles that hold one;
//  * 16-byte global accesses: the vector-memory pipe takes a wave's addresses four lanes a cycle whatever the width, and
//    hpss_tile_kernel's 35 dword loads + 64 dword stores per thread cost it ~6 000 cycles per tile (0.25 ms of the 0.92).
//    The fill reads whole 16-byte chunks wherever a chunk lies inside the row (edge chunks: four reflected scalar loads);
//    results leave as 16-byte stores (4-byte aligned: rows of 1025 bins start anywhere) — 9 + 16 instructions per thread;
//  * two thread -> output maps: along B (the fast axis, 16-byte LDS reads) a thread owns 2 rows x 8 consecutive columns, where the
//    masks are formed too; along A it owns one column x 16 rows (two runs; LDS reads down the rows, a wave = 64 consecutive
//    columns), and the A medians cross the workgroup once through the (by then dead) tile;
//  * the results leave through a per-wave transpose in LDS (hp_emit_block): four whole 256-byte row segments per store instruction.
#ifndef TAC_HPSS_OCC8
#define TAC_HPSS_OCC8 3
#endif
#ifndef TAC_HPSS_STRIDE8
#define TAC_HPSS_STRIDE8 96    // (100 / 104: 4 % slower, tools/ablation/README.md)
#endif
constexpr int HP8_STRIDE = TAC_HPSS_STRIDE8;  // floats per tile row
constexpr int HP8_LDS_FLOATS = HP_ROWS * HP8_STRIDE;
constexpr int HP8_EX = 68;                    // row stride of the 64 x 64 exchange of the A medians
typedef float hp_f4u __attribute__((ext_vector_type(4), aligned(4)));

// n_rows x COLS floats (COLS a multiple of 4) of a tile, from rows a_first .. and columns b_first .. of one spectrogram,
// both reflected at the borders (positions only out-of-range outputs would use are clamped).  Returns "this thread saw a NaN".
template <int COLS>
__device__ __forceinline__ bool hp_fill(float* tile, int stride, const float* __restrict__ xr, int a_first, int n_rows,
                                        int b_first, int NA, int NB, long long sa, long long sb, int tid) {
    bool seen_nan = false;
    if (sb == 1) {
        constexpr int CH = COLS / 4;
        for (int q = tid; q < n_rows * CH; q += 256) {
            const int r = q / CH, c4 = (q - r * CH) * 4;
            const float* src = xr + (long long)reflect_clamped(a_first + r, NA) * sa;
            const int b = b_first + c4;
            hp_f4 v;
            if (b >= 0 && b + 3 < NB) {
                const hp_f4u u = *reinterpret_cast<const hp_f4u*>(src + b);
                v.x = u.x; v.y = u.y; v.z = u.z; v.w = u.w;
            } else {
                v.x = src[reflect_clamped(b, NB)];
                v.y = src[reflect_clamped(b + 1, NB)];
                v.z = src[reflect_clamped(b + 2, NB)];
                v.w = src[reflect_clamped(b + 3, NB)];
            }
            seen_nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
            *reinterpret_cast<hp_f4*>(tile + r * stride + c4) = v;
        }
        return seen_nan;
    }
    for (int i = tid; i < n_rows * COLS; i += 256) {
        const int r = i / COLS, c = i - r * COLS;
        const int a = reflect_clamped(a_first + r, NA), b = reflect_clamped(b_first + c, NB);
        const float v = xr[(long long)a * sa + (long long)b * sb];
        seen_nan |= v != v;
        tile[r * stride + c] = v;
    }
    return seen_nan;
}

// eight windows of K taps with torch.median's NaN rule: a window that holds a NaN has a NaN median (the min / max network
// alone would drop it); the bookkeeping runs only in tiles that hold one (workgroup-uniform flag, almost never set)
template <int K>
__device__ __forceinline__ void hp_run8(const float (&w)[K + 7], float (&med)[8], int tile_has_nan) {
    median_run8_any<K>(w, med);
    if (tile_has_nan) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bool bad = false;
#pragma unroll
            for (int u = 0; u < K; ++u) bad |= w[j + u] != w[j + u];
            med[j] = bad ? __builtin_nanf("") : med[j];
        }
    }
}

// masks and the four results of eight consecutive columns bq .. bq + 7 of row a (o = the offset of (a, bq))
__device__ __forceinline__ void hp_emit8(bool row_ok, int bq, int NB, long long o, long long sb, const float (&m_a)[8],
                                         const float (&m_b)[8], const float (&centre)[8], int b_is_time, float power,
                                         int hard, float* harm_o, float* perc_o, float* mh_o, float* mp_o) {
    float mh[8], mp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float harm = b_is_time ? m_b[j] : m_a[j];
        const float perc = b_is_time ? m_a[j] : m_b[j];
        hpss_masks(harm, perc, power, hard, mh[j], mp[j]);
    }
    if (!row_ok) return;
    auto put = [&](float* base, const float (&v)[8]) {
        if (sb == 1 && bq + 7 < NB) {
            hp_f4u lo4, hi4;
            lo4.x = v[0]; lo4.y = v[1]; lo4.z = v[2]; lo4.w = v[3];
            hi4.x = v[4]; hi4.y = v[5]; hi4.z = v[6]; hi4.w = v[7];
            *reinterpret_cast<hp_f4u*>(base + o) = lo4;
            *reinterpret_cast<hp_f4u*>(base + o + 4) = hi4;
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (bq + j < NB) base[o + (long long)j * sb] = v[j];
    };
    put(mh_o, mh);
    put(mp_o, mp);
    if (harm_o) {
        float hv[8], pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hv[j] = centre[j] * mh[j];
            pv[j] = centre[j] * mp[j];
        }
        put(harm_o, hv);
        put(perc_o, pv);
    }
}

// The same for the thread's 2 rows x 8 columns of the B map, stored through a per-wave transpose in LDS: in the B map a
// store instruction's lanes write 16-byte pieces 32 bytes apart (the other half of each lane's eight columns goes with the next
// instruction), which the memory pipe cannot merge — one request per lane; staged as [16 rows][64 columns] (HP8_EX floats apart) and read
// back with lane l on row l / 16 (+ 4 k), columns 4 (l % 16) .., every store instruction writes four whole 256-byte row segments.
// `scratch`: this wave's 16 x HP8_EX floats; a_blk: first row of the wave's 16-row block (rows 2 (ay % 8) + i of it are this thread's).
constexpr int HP8_SCRATCH = 16 * 68;          // floats per wave (64-column tiles)
// TB: columns of the tile (64: a wave's block is 16 rows x 64 columns; 128: 8 x 128; 256: 4 x 256 — a store instruction then writes
// 1 KB of one row)
template <int TB = 64>
__device__ __forceinline__ void hp_emit_block(float* scratch, int tid, int a_blk, int b0, int NA, int NB, long long row_off,
                                              long long sa, long long sb, const float (&m_a)[2][8], const float (&m_b)[2][8],
                                              const float (&centre)[2][8], int b_is_time, float power, int hard, float* harm_o,
                                              float* perc_o, float* mh_o, float* mp_o) {
    constexpr int NBX = TB / 8, PITCH = TB + 4, CPR = TB / 4;
    const int lane = tid & 63, bx = lane % NBX, lr = 2 * (lane / NBX);
    if (sb == 1) {
        float mh[2][8], mp[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float harm = b_is_time ? m_b[i][j] : m_a[i][j];
                const float perc = b_is_time ? m_a[i][j] : m_b[i][j];
                hpss_masks(harm, perc, power, hard, mh[i][j], mp[i][j]);
            }
        auto put = [&](float* base, const float (&v)[2][8]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                hp_f4 lo4, hi4;
                lo4.x = v[i][0]; lo4.y = v[i][1]; lo4.z = v[i][2]; lo4.w = v[i][3];
                hi4.x = v[i][4]; hi4.y = v[i][5]; hi4.z = v[i][6]; hi4.w = v[i][7];
                hp_f4* dst = reinterpret_cast<hp_f4*>(scratch + (lr + i) * PITCH + 8 * bx);
                dst[0] = lo4;
                dst[1] = hi4;
            }
            wave_lds_fence();               // (wave-synchronous transpose: every lane's slots are written before any lane reads)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k, r = c / CPR, sc = 4 * (c % CPR);
                const hp_f4 x4 = *reinterpret_cast<const hp_f4*>(scratch + r * PITCH + sc);
                const int a = a_blk + r, b = b0 + sc;
                if (a < NA) {
                    float* dst = base + row_off + (long long)a * sa + b;
                    if (b + 3 < NB) {
                        hp_f4u u;
                        u.x = x4.x; u.y = x4.y; u.z = x4.z; u.w = x4.w;
                        *reinterpret_cast<hp_f4u*>(dst) = u;
                    } else {
                        if (b < NB) dst[0] = x4.x;
                        if (b + 1 < NB) dst[1] = x4.y;
                        if (b + 2 < NB) dst[2] = x4.z;
                    }
                }
            }
            wave_lds_fence();               // ... and read back before the next array's values overwrite them
        };
        put(mh_o, mh);
        put(mp_o, mp);
        if (harm_o) {
            float hv[2][8], pv[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    hv[i][j] = centre[i][j] * mh[i][j];
                    pv[i][j] = centre[i][j] * mp[i][j];
                }
            put(harm_o, hv);
            put(perc_o, pv);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int a = a_blk + lr + i, bq = b0 + 8 * bx;
        hp_emit8(a < NA, bq, NB, row_off + (long long)a * sa + (long long)bq * sb, sb, m_a[i], m_b[i], centre[i], b_is_time, power,
                 hard, harm_o, perc_o, mh_o, mp_o);
    }
}

// K + 7 taps of one tile row starting at column `first` (any alignment), read as 16-byte chunks
template <int K, int LEFT = HP_LEFT>
__device__ __forceinline__ void hp_row_taps(const float* tile_row, int first_aligned, float (&w)[K + 7]) {
    constexpr int OFF = (LEFT - K / 2) & 3, NCH = (K + 7 + OFF + 3) / 4;
    const hp_f4* src = reinterpret_cast<const hp_f4*>(tile_row + first_aligned);
    float buf[4 * NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const hp_f4 v = src[u];
        buf[4 * u] = v.x; buf[4 * u + 1] = v.y; buf[4 * u + 2] = v.z; buf[4 * u + 3] = v.w;
    }
#pragma unroll
    for (int u = 0; u < K + 7; ++u) w[u] = buf[OFF + u];
}

template <int K>
__global__ void __launch_bounds__(256, TAC_HPSS_OCC8)
hpss_tile8_kernel(const float* __restrict__ x, int NA, int NB, long long sr, long long sa, long long sb, int tiles_a,
                  int tiles_b, long long total_tiles, int b_is_time, float power, int hard, float* __restrict__ harm_o, float* __restrict__ perc_o,
                  float* __restrict__ mh_o, float* __restrict__ mp_o) {
    constexpr int HALF = K / 2;
    __shared__ __attribute__((aligned(16))) float tile[HP8_LDS_FLOATS];
    const int tid = threadIdx.x;
    const int per_row = tiles_a * tiles_b;
    const long long tile_id = hp_tile_of_block(total_tiles);
    if (tile_id >= total_tiles) return;                     // (uniform: the grid is rounded up to a multiple of eight)
    const long long row = tile_id / per_row;
    const int rem = (int)(tile_id - row * per_row);
    const int a0 = (rem / tiles_b) * HP_TILE, b0 = (rem % tiles_b) * HP_TILE;
    const float* xr = x + row * sr;
    const bool seen_nan = hp_fill<HP_STRIDE>(tile, HP8_STRIDE, xr, a0 - 15, HP_ROWS, b0 - HP_LEFT, NA, NB, sa, sb, tid);
    const int tile_has_nan = __syncthreads_or(seen_nan ? 1 : 0);
    HP_PROBE_BEGIN();
    // ---- along B: thread (bx, ay) owns rows 2 ay, 2 ay + 1 and columns 8 bx .. 8 bx + 7
    const int bx = tid & 7, ay = tid >> 3;
    constexpr int START = HP_LEFT - HALF, OFF = START & 3;
    float medB[2][8], centre[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float w[K + 7];
        hp_row_taps<K>(tile + (2 * ay + 15 + i) * HP8_STRIDE, 8 * bx + (START - OFF), w);
#pragma unroll
        for (int j = 0; j < 8; ++j) centre[i][j] = w[HALF + j];
        hp_run8<K>(w, medB[i], tile_has_nan);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- along A: thread (cx, ry) owns column cx and rows 16 ry .. 16 ry + 15 (two runs)
    const int cx = tid & 63, ry = tid >> 6;
    float medA[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float w0[K + 7];
        const float* src = tile + (16 * ry + 8 * h + 15 - HALF) * HP8_STRIDE + cx + HP_LEFT;
#pragma unroll
        for (int u = 0; u < K + 7; ++u) w0[u] = src[u * HP8_STRIDE];
        hp_run8<K>(w0, medA[h], tile_has_nan);
        __builtin_amdgcn_sched_barrier(0);
    }
    HP_PROBE_END();
    __syncthreads();                         // every read of the tile is done: the A medians change maps through it
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[(16 * ry + 8 * h + i) * HP8_EX + cx] = medA[h][i];
    __syncthreads();
    // ---- masks and stores: the B map's values, stored through the wave's transpose area (behind the exchange rows, in the dead tile)
    float m_a[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const hp_f4 ma0 = *reinterpret_cast<const hp_f4*>(tile + (2 * ay + i) * HP8_EX + 8 * bx);
        const hp_f4 ma1 = *reinterpret_cast<const hp_f4*>(tile + (2 * ay + i) * HP8_EX + 8 * bx + 4);
        m_a[i][0] = ma0.x; m_a[i][1] = ma0.y; m_a[i][2] = ma0.z; m_a[i][3] = ma0.w;
        m_a[i][4] = ma1.x; m_a[i][5] = ma1.y; m_a[i][6] = ma1.z; m_a[i][7] = ma1.w;
    }
    static_assert(HP_TILE * HP8_EX + 4 * HP8_SCRATCH <= HP8_LDS_FLOATS, "exchange rows + four transpose areas fit the tile");
    hp_emit_block(tile + HP_TILE * HP8_EX + (tid >> 6) * HP8_SCRATCH, tid, a0 + 16 * (tid >> 6), b0, NA, NB, row * sr, sa, sb, m_a, medB,
                  centre, b_is_time, power, hard, harm_o, perc_o, mh_o, mp_o);
}

// hpss_tilew_kernel<K, TA, TB>: hpss_tile8_kernel on TA x TB tiles (TA * TB = 4096 outputs, TB a multiple of 64 columns of the fast axis).
// The four result arrays leave as TB * 4-byte row segments; the store pattern ALONE (tools/ubench/tile_store_rate.hip) runs at 3.95 TB/s
// with 256-byte segments, 4.58 with 512, 5.07 with 1 024 — 64 x 64 tiles spend 0.33 ms just writing.  Same maps: along B a thread owns
// 2 rows x 8 columns, along A one column x 16 rows; fast axis contiguous only.
template <int K, int TA, int TB>
__global__ void __launch_bounds__(256, TAC_HPSS_OCC8)
hpss_tilew_kernel(const float* __restrict__ x, int NA, int NB, long long sr, long long sa, int tiles_a, int tiles_b,
                  long long total_tiles, int b_is_time, float power, int hard, float* __restrict__ harm_o,
                  float* __restrict__ perc_o, float* __restrict__ mh_o, float* __restrict__ mp_o) {
    static_assert(TA * TB == 4096 && TA % 16 == 0 && TB % 64 == 0, "256 threads x 16 outputs");
    constexpr int HALF = K / 2, ROWS = TA + 30, STRIDE = HP_LEFT + TB + 16, EX = TB + 4, NBX = TB / 8, ROWS_W = 128 / NBX;
    static_assert(TA * EX + 4 * ROWS_W * EX <= ROWS * STRIDE, "exchange rows + four transpose areas fit the tile");
    __shared__ __attribute__((aligned(16))) float tile[ROWS * STRIDE];
    const int tid = threadIdx.x;
    const int per_row = tiles_a * tiles_b;
    const long long tile_id = hp_tile_of_block(total_tiles);
    if (tile_id >= total_tiles) return;
    const long long row = tile_id / per_row;
    const int rem = (int)(tile_id - row * per_row);
    const int a0 = (rem / tiles_b) * TA, b0 = (rem % tiles_b) * TB;
    const bool seen_nan = hp_fill<STRIDE>(tile, STRIDE, x + row * sr, a0 - 15, ROWS, b0 - HP_LEFT, NA, NB, sa, 1, tid);
    const int tile_has_nan = __syncthreads_or(seen_nan ? 1 : 0);
    const int bx = tid % NBX, ay = tid / NBX;
    constexpr int START = HP_LEFT - HALF, OFF = START & 3;
    float medB[2][8], centre[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float w[K + 7];
        hp_row_taps<K>(tile + (2 * ay + 15 + i) * STRIDE, 8 * bx + (START - OFF), w);
#pragma unroll
        for (int j = 0; j < 8; ++j) centre[i][j] = w[HALF + j];
        hp_run8<K>(w, medB[i], tile_has_nan);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int cx = tid % TB, ry = tid / TB;
    float medA[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float w0[K + 7];
        const float* src = tile + (16 * ry + 8 * h + 15 - HALF) * STRIDE + cx + HP_LEFT;
#pragma unroll
        for (int u = 0; u < K + 7; ++u) w0[u] = src[u * STRIDE];
        hp_run8<K>(w0, medA[h], tile_has_nan);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                         // every read of the tile is done: the A medians change maps through it
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[(16 * ry + 8 * h + i) * EX + cx] = medA[h][i];
    __syncthreads();
    float m_a[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const hp_f4 ma0 = *reinterpret_cast<const hp_f4*>(tile + (2 * ay + i) * EX + 8 * bx);
        const hp_f4 ma1 = *reinterpret_cast<const hp_f4*>(tile + (2 * ay + i) * EX + 8 * bx + 4);
        m_a[i][0] = ma0.x; m_a[i][1] = ma0.y; m_a[i][2] = ma0.z; m_a[i][3] = ma0.w;
        m_a[i][4] = ma1.x; m_a[i][5] = ma1.y; m_a[i][6] = ma1.z; m_a[i][7] = ma1.w;
    }
    hp_emit_block<TB>(tile + TA * EX + (tid >> 6) * (ROWS_W * EX), tid, a0 + ROWS_W * (tid >> 6), b0, NA, NB, row * sr, sa, 1, m_a, medB,
                      centre, b_is_time, power, hard, harm_o, perc_o, mh_o, mp_o);
}

// (hpss_tile8p_kernel — the same as persistent workgroups that request the next tile before storing the current one: +5 % at K = 31,
// -1 % at K = 9, tools/ablation/README.md — lives in tools/ablation/lab_knobs_r06.patch.)

// Unequal (or small) widths, round 4: two launches over the same 64 x 64 output tiles, each with a halo along ONE axis.
// hpss_axis_a_kernel<KA>: medians along A (the slow memory axis) into `tmp` — the caller's mask_perc buffer, no workspace.
// hpss_axis_b_kernel<KB>: medians along B, the A medians read back from `tmp` (a wave reads exactly the 16 x 64 block it then
// overwrites, and has waited for those loads — they feed the masks — before its first store), masks and the four results.  (Round 3's one-thread-per-element kernel issued 64 dword loads per output and
// ran at 7 % of the HBM peak; hpss_kernel below is kept as the A/B baseline and for non-unit fast strides.)
template <int K>
__global__ void __launch_bounds__(256)
hpss_axis_a_kernel(const float* __restrict__ x, int NA, int NB, long long sr, long long sa, long long sb, int tiles_a,
                   int tiles_b, long long total_tiles, float* __restrict__ tmp) {
    constexpr int HALF = K / 2, ROWS = HP_TILE + K - 1;
    __shared__ __attribute__((aligned(16))) float tile[ROWS * HP_TILE];
    const int tid = threadIdx.x;
    const int per_row = tiles_a * tiles_b;
    const long long tile_id = hp_tile_of_block(total_tiles);
    if (tile_id >= total_tiles) return;                     // (uniform: the grid is rounded up to a multiple of eight)
    const long long row = tile_id / per_row;
    const int rem = (int)(tile_id - row * per_row);
    const int a0 = (rem / tiles_b) * HP_TILE, b0 = (rem % tiles_b) * HP_TILE;
    const bool seen_nan = hp_fill<HP_TILE>(tile, HP_TILE, x + row * sr, a0 - HALF, ROWS, b0, NA, NB, sa, sb, tid);
    const int tile_has_nan = __syncthreads_or(seen_nan ? 1 : 0);
    const int cx = tid & 63, ry = tid >> 6;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float w0[K + 7], med[8];
        const float* src = tile + (16 * ry + 8 * h) * HP_TILE + cx;
#pragma unroll
        for (int u = 0; u < K + 7; ++u) w0[u] = src[u * HP_TILE];          // (row 16 ry + 8 h + u <= 62 + K = ROWS - 1)
        hp_run8<K>(w0, med, tile_has_nan);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = a0 + 16 * ry + 8 * h + i, b = b0 + cx;
            if (a < NA && b < NB) tmp[row * sr + (long long)a * sa + (long long)b * sb] = med[i];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int K>
__global__ void __launch_bounds__(256)
hpss_axis_b_kernel(const float* __restrict__ x, int NA, int NB, long long sr, long long sa, long long sb, int tiles_a,
                   int tiles_b, long long total_tiles, int b_is_time, float power, int hard, float* harm_o, float* perc_o, float* mh_o,
                   float* mp_o /* holds the A medians on entry */) {
    // widths above 31 (round 5): a 32-column low-side halo and 128-float rows (32 + 64 + 31 = 127); the rest is unchanged
    constexpr int HALF = K / 2, LEFT = K <= 31 ? HP_LEFT : 32, STRIDE = K <= 31 ? HP_STRIDE : 128;
    static_assert(HALF <= LEFT && LEFT + HP_TILE + HALF <= STRIDE, "halo does not fit the tile row");
    __shared__ __attribute__((aligned(16))) float tile[HP_TILE * STRIDE + 4 * HP8_SCRATCH];   // + the waves' store-transpose areas
    const int tid = threadIdx.x;
    const int per_row = tiles_a * tiles_b;
    const long long tile_id = hp_tile_of_block(total_tiles);
    if (tile_id >= total_tiles) return;                     // (uniform: the grid is rounded up to a multiple of eight)
    const long long row = tile_id / per_row;
    const int rem = (int)(tile_id - row * per_row);
    const int a0 = (rem / tiles_b) * HP_TILE, b0 = (rem % tiles_b) * HP_TILE;
    const bool seen_nan = hp_fill<STRIDE>(tile, STRIDE, x + row * sr, a0, HP_TILE, b0 - LEFT, NA, NB, sa, sb, tid);
    const int tile_has_nan = __syncthreads_or(seen_nan ? 1 : 0);
    const int bx = tid & 7, ay = tid >> 3;
    constexpr int START = LEFT - HALF, OFF = START & 3;
    float medB[2][8], centre[2][8], m_a[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float w[K + 7];
        hp_row_taps<K, LEFT>(tile + (2 * ay + i) * STRIDE, 8 * bx + (START - OFF), w);
#pragma unroll
        for (int j = 0; j < 8; ++j) centre[i][j] = w[HALF + j];
        hp_run8<K>(w, medB[i], tile_has_nan);
        // the A medians of these sixteen elements: read by the wave that overwrites them below (its 16 x 64 block), and waited
        // for — they feed the masks — before any of its stores is issued
        const int a = a0 + 2 * ay + i, bq = b0 + 8 * bx;
        const long long o = row * sr + (long long)a * sa + (long long)bq * sb;
        if (a < NA && sb == 1 && bq + 7 < NB) {
            const hp_f4u lo4 = *reinterpret_cast<const hp_f4u*>(mp_o + o), hi4 = *reinterpret_cast<const hp_f4u*>(mp_o + o + 4);
            m_a[i][0] = lo4.x; m_a[i][1] = lo4.y; m_a[i][2] = lo4.z; m_a[i][3] = lo4.w;
            m_a[i][4] = hi4.x; m_a[i][5] = hi4.y; m_a[i][6] = hi4.z; m_a[i][7] = hi4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) m_a[i][j] = (a < NA && bq + j < NB) ? mp_o[o + (long long)j * sb] : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    hp_emit_block(tile + HP_TILE * STRIDE + (tid >> 6) * HP8_SCRATCH, tid, a0 + 16 * (tid >> 6), b0, NA, NB, row * sr, sa, sb, m_a, medB,
                  centre, b_is_time, power, hard, harm_o, perc_o, mh_o, mp_o);
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
    const long long total_tiles = rows * (long long)ta * tb;
    hipLaunchKernelGGL(hpss_tile8_kernel<K>, dim3(hp_grid_for(total_tiles)), dim3(256), 0, stream, mag, NA, NB, sr, sa, sb, ta, tb,
                       total_tiles, t_fast ? 1 : 0, power, hard, harm, perc, mh, mp);
}

}  // namespace tac

namespace tac {

// ---------------------------------------------------------------- gradient (round 6)
// d hpss / d mag (beta_hpss.py:35-127 under autograd).  With H / P the medians along time / frequency of the reflect-padded spectrogram,
// Hp = H^power, Pp = P^power, D = Hp + Pp + eps:  harm = mag mh, perc = mag mp, mh = (Hp + eps) / D, mp = (Pp + eps) / D (soft masks).
//   direct:  g_mag += g_harm mh + g_perc mp
//   masks:   G_h = g_mh + g_harm mag, G_p = g_mp + g_perc mag;  g_Hp = (G_h Pp - G_p (Pp + eps)) / D^2,  g_Pp = (G_p Hp - G_h (Hp + eps)) / D^2
//   medians: g_H = g_Hp power H^(power-1) goes to THE element the time window's median selected, g_P to the frequency window's
// (torch.median's gradient; hard masks are not differentiable: only the direct term).  One thread per element re-selects both medians by
// rank counting (the element with exactly (k-1)/2 smaller ones, ties by position) and scatters with float atomics: the sum order, hence
// the last bits, vary from run to run.  Training through hpss is not on the measured path: nothing here is tuned (k^2 / 2 compares per window).
__device__ __forceinline__ int hp_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__global__ void __launch_bounds__(256)
hpss_backward_kernel(const float* __restrict__ mag, long long rows, int n_freqs, int n_frames, long long stride_r, long long stride_f,
                     long long stride_t, int kf, int kt, float power, int hard, const float* __restrict__ g_harm,
                     const float* __restrict__ g_perc, const float* __restrict__ g_mh, const float* __restrict__ g_mp,
                     float* __restrict__ g_mag) {
    const long long total = rows * n_freqs * (long long)n_frames;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        // consecutive threads walk the faster-varying axis of the layout
        const bool t_fast = stride_t <= stride_f;
        const int nb = t_fast ? n_frames : n_freqs;
        const long long q = e / nb;
        const int ib = (int)(e - q * nb), ia = (int)(q % (t_fast ? n_freqs : n_frames));
        const long long r = q / (t_fast ? n_freqs : n_frames);
        const int f = t_fast ? ia : ib, t = t_fast ? ib : ia;
        const float* base = mag + r * stride_r;
        const long long at = r * stride_r + (long long)f * stride_f + (long long)t * stride_t;
        // median of `k` values along one axis (stride `st`, length `n`, centre `c`): value and source index
        auto median = [&](const float* line, long long st, int n, int c, int k, int& src) -> float {
            const int half = k / 2;
            for (int i = 0; i < k; ++i) {
                const int si = hp_reflect(c - half + i, n);
                const float v = line[(long long)si * st];
                int below = 0;
                for (int j = 0; j < k; ++j) {
                    const float u = line[(long long)hp_reflect(c - half + j, n) * st];
                    below += (u < v || (u == v && j < i)) ? 1 : 0;
                }
                if (below == half) {
                    src = si;
                    return v;
                }
            }
            src = c;                                   // (NaNs in the window: no rank matches; the gradient stays where it is)
            return line[(long long)c * st];
        };
        int src_t = t, src_f = f;
        const float H = median(base + (long long)f * stride_f, stride_t, n_frames, t, kt, src_t);
        const float P = median(base + (long long)t * stride_t, stride_f, n_freqs, f, kf, src_f);
        const float x = mag[at];
        const float Hp = power == 1.0f ? H : powf(H, power), Pp = power == 1.0f ? P : powf(P, power);
        const float gh = g_harm ? g_harm[at] : 0.0f, gp = g_perc ? g_perc[at] : 0.0f;
        float direct;
        if (hard) {
            direct = gh * (Hp > Pp ? 1.0f : 0.0f) + gp * (Hp < Pp ? 1.0f : 0.0f);
        } else {
            const float eps = 1e-6f, D = Hp + Pp + eps, inv = 1.0f / D;
            direct = gh * ((Hp + eps) * inv) + gp * ((Pp + eps) * inv);
            const float Gh = (g_mh ? g_mh[at] : 0.0f) + gh * x, Gp = (g_mp ? g_mp[at] : 0.0f) + gp * x;
            float gHp = (Gh * Pp - Gp * (Pp + eps)) * inv * inv, gPp = (Gp * Hp - Gh * (Hp + eps)) * inv * inv;
            if (power != 1.0f) {
                gHp *= power * powf(H, power - 1.0f);
                gPp *= power * powf(P, power - 1.0f);
            }
            if (gHp != 0.0f) unsafeAtomicAdd(g_mag + r * stride_r + (long long)f * stride_f + (long long)src_t * stride_t, gHp);
            if (gPp != 0.0f) unsafeAtomicAdd(g_mag + r * stride_r + (long long)src_f * stride_f + (long long)t * stride_t, gPp);
        }
        if (direct != 0.0f) unsafeAtomicAdd(g_mag + at, direct);
    }
}

}  // namespace tac

extern "C" {

int tac_hpss_backward_f32(const float* mag, int64_t rows, int32_t n_freqs, int32_t n_frames, int64_t stride_r, int64_t stride_f,
                          int64_t stride_t, int32_t kernel_f, int32_t kernel_t, float power, int hard, const float* grad_harm,
                          const float* grad_perc, const float* grad_mask_harm, const float* grad_mask_perc, float* grad_mag,
                          void* stream) {
    using namespace tac;
    if (rows == 0 || n_freqs == 0 || n_frames == 0) return TAC_OK;
    if (!mag || !grad_mag || rows < 0 || n_freqs < 0 || n_frames < 0 || stride_r < 0 || stride_f < 0 || stride_t < 0) return TAC_E_INVALID;
    if (kernel_f < 1 || kernel_t < 1 || !(kernel_f & 1) || !(kernel_t & 1) || kernel_f > 63 || kernel_t > 63)
        return TAC_E_UNSUPPORTED;
    if (kernel_f / 2 >= n_freqs || kernel_t / 2 >= n_frames) return TAC_E_SHORT_INPUT;
    const long long total = rows * n_freqs * (long long)n_frames;
    long long blocks = (total + 255) / 256;
    const long long cap = (long long)device_cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(hpss_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mag, (long long)rows, (int)n_freqs,
                       (int)n_frames, (long long)stride_r, (long long)stride_f, (long long)stride_t, (int)kernel_f, (int)kernel_t, power, hard,
                       grad_harm, grad_perc, grad_mask_harm, grad_mask_perc, grad_mag);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_hpss_f32(const float* mag, int64_t rows, int32_t n_freqs, int32_t n_frames, int64_t stride_r, int64_t stride_f,
                 int64_t stride_t, int32_t kernel_f, int32_t kernel_t, float power, int hard, float* harm, float* perc,
                 float* mask_harm, float* mask_perc, void* stream) {
    using namespace tac;
    if (rows == 0 || n_freqs == 0 || n_frames == 0) return TAC_OK;
    if (!mag || !mask_harm || !mask_perc || (harm == nullptr) != (perc == nullptr)) return TAC_E_INVALID;
    if (rows < 0 || n_freqs < 0 || n_frames < 0) return TAC_E_INVALID;
    if (kernel_f < 1 || kernel_t < 1 || !(kernel_f & 1) || !(kernel_t & 1) || kernel_f > 63 || kernel_t > 63)
        return TAC_E_UNSUPPORTED;
    if (kernel_f / 2 >= n_freqs || kernel_t / 2 >= n_frames) return TAC_E_SHORT_INPUT;      // reflect padding needs pad < size
    {
        // the outputs must not overlap the input or one another (header, (10)): every kernel reads halos of mag while other
        // workgroups store, and the two-launch route parks its first medians in mask_perc.  Checked on the address ranges the
        // strides span (non-negative strides; anything else is left to the caller).
        if (stride_r >= 0 && stride_f >= 0 && stride_t >= 0) {
            const unsigned long long span = 4ull * (unsigned long long)((rows - 1) * stride_r + (long long)(n_freqs - 1) * stride_f +
                                                                        (long long)(n_frames - 1) * stride_t + 1);
            const void* ptrs[5] = {mag, mask_harm, mask_perc, harm, perc};
            for (int i = 0; i < 5; ++i)
                for (int j = i + 1; j < 5; ++j) {
                    if (!ptrs[i] || !ptrs[j]) continue;
                    const unsigned long long a = reinterpret_cast<unsigned long long>(ptrs[i]), b = reinterpret_cast<unsigned long long>(ptrs[j]);
                    if (a < b + span && b < a + span) return TAC_E_INVALID;
                }
        }
    }
    const long long tiles = rows * ((n_freqs + HP_TILE - 1) / HP_TILE) * ((n_frames + HP_TILE - 1) / HP_TILE);
    if (kernel_f == kernel_t && kernel_f >= 9 && kernel_f <= 31 && tiles < 0x7fffffffLL) {
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
    if (tiles < 0x7fffffffLL) {
        const bool t_fast = stride_t <= stride_f;
        const int NA = t_fast ? n_freqs : n_frames, NB = t_fast ? n_frames : n_freqs;
        const long long sa = t_fast ? stride_f : stride_t, sb = t_fast ? stride_t : stride_f;
        const int ka = t_fast ? kernel_f : kernel_t, kb = t_fast ? kernel_t : kernel_f;
        const int ta = (NA + HP_TILE - 1) / HP_TILE, tb = (NB + HP_TILE - 1) / HP_TILE;
        const dim3 grid(hp_grid_for(tiles));
#define TAC_HPSS_ODD(M) M(1) M(3) M(5) M(7) M(9) M(11) M(13) M(15) M(17) M(19) M(21) M(23) M(25) M(27) M(29) M(31)            \
    M(33) M(35) M(37) M(39) M(41) M(43) M(45) M(47) M(49) M(51) M(53) M(55) M(57) M(59) M(61) M(63)
#define TAC_HPSS_A(K)                                                                                                     \
    case K:                                                                                                               \
        hipLaunchKernelGGL(hpss_axis_a_kernel<K>, grid, dim3(256), 0, (hipStream_t)stream, mag, NA, NB,                   \
                           (long long)stride_r, sa, sb, ta, tb, (long long)tiles, mask_perc);                             \
        break;
#define TAC_HPSS_B(K)                                                                                                     \
    case K:                                                                                                               \
        hipLaunchKernelGGL(hpss_axis_b_kernel<K>, grid, dim3(256), 0, (hipStream_t)stream, mag, NA, NB,                   \
                           (long long)stride_r, sa, sb, ta, tb, (long long)tiles, t_fast ? 1 : 0, power, hard, harm, perc, mask_harm, \
                           mask_perc);                                                                                    \
        break;
        switch (ka) { TAC_HPSS_ODD(TAC_HPSS_A) }
        TAC_HIP(hipGetLastError());
        switch (kb) { TAC_HPSS_ODD(TAC_HPSS_B) }
#undef TAC_HPSS_A
#undef TAC_HPSS_B
#undef TAC_HPSS_ODD
        TAC_HIP(hipGetLastError());
        return TAC_OK;
    }
    if (kernel_f > 32 || kernel_t > 32) return TAC_E_UNSUPPORTED;      // (the one-thread-per-element form sorts 32 taps)
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
