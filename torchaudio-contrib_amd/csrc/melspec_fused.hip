// melspec_fused.hip — STFT -> |.|^p -> mel filterbank (fp32 MFMA) -> dB in ONE kernel.
//
// Fuses the whole Sequential(*Melspectrogram(...), AmplitudeToDb()) chain of the reference
// (layers.py:307-381; functional.py:99-107, 126-128, 183-184, 291-296) so that neither the complex
// STFT (657 MB at cfg-2), the power spectrogram (328 MB) nor the linear mel tensor ever touch HBM:
// algorithmic traffic is 4*hop bytes in + 4*M bytes out per frame.
//
// Per workgroup (8 waves, one CU): loop over tiles of 16 consecutive frames of one row.
//   phase A  each wave FFTs its frames (fft_core.hpp) and writes |X|^p rows into the P tile in LDS
//   phase B  P[16 x F] · fb[F x M] on v_mfma_f32_16x16x4_f32.  The triangular filters make fb
//            block-sparse: for every 16-band tile only the bins [klo, khi) from the plan carry
//            weight, so the K loop of a tile covers just that range.  The K-steps of all tiles are
//            cut into 8 equal contiguous shares (one per wave) so the wide high-frequency tiles do
//            not serialise on one SIMD; shares write partial 16x16 tiles to LDS slots in a fixed order
//   phase C  fixed-order sum of a tile's partials (deterministic), optional dB epilogue, coalesced
//            512-byte row stores of out[row][frame][0..M)
#include "host_common.hpp"

namespace tac {

constexpr int MEL_WAVES = 8;
constexpr int MEL_TILE = 16;                 // frames per tile = MFMA M dimension
constexpr int MEL_MAX_BAND_TILES = 32;       // n_mels <= 512
constexpr int MEL_MAX_SLOTS = MEL_MAX_BAND_TILES + MEL_WAVES;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MelArgs {
    const float* fb;       // [F][M]
    const int* plan;       // [2*ntiles]
    int n_mels;
    int n_band_tiles;
    int pstride;           // floats per P-tile row
    float power;
    int db;
    float amin;
    float log10_ref;
    float* out;            // [rows][T][M]
};

struct SegTable {
    int nslots;
    int seg_tile[MEL_MAX_SLOTS];
    int seg_k0[MEL_MAX_SLOTS];
    int seg_steps[MEL_MAX_SLOTS];
    int seg_wave[MEL_MAX_SLOTS];
    int tile_first[MEL_MAX_BAND_TILES];
    int tile_count[MEL_MAX_BAND_TILES];
};

template <int NC>
__host__ __device__ constexpr int mel_pstride() {
    // >= F + 3 (K-steps may overrun the last bin by 3) and == 2 (mod 32): the MFMA A-operand read
    // P[frame = lane&15][k0 + (lane>>4)] then hits 32 distinct banks per 32-lane group.
    int need = NC + 1 + 3;
    int s = (need / 32) * 32 + 2;
    while (s < need) s += 32;
    return s;
}

template <int NC, int E>
__global__ void __launch_bounds__(MEL_WAVES * 64)
melspec_kernel(FrameGeom g, Tables tb, MelArgs m) {
    using F = WaveFft<NC, E>;
    constexpr int NBINS = NC + 1;
    constexpr int PSTRIDE = mel_pstride<NC>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* scratch = reinterpret_cast<cf*>(smem_raw);                                  // MEL_WAVES*G*PADDED
    float* ptile = reinterpret_cast<float*>(scratch + MEL_WAVES * F::G * F::PADDED);  // MEL_TILE*PSTRIDE
    float* partial = ptile + MEL_TILE * PSTRIDE;                                    // nslots*256
    SegTable* seg = reinterpret_cast<SegTable*>(partial + (m.n_band_tiles + MEL_WAVES) * 256);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int sub = lane / F::LPF;
    const int t = lane % F::LPF;
    cf* lds = scratch + (w * F::G + sub) * F::PADDED;

    // ---- one-off setup: K-step shares, zero the P-tile padding columns
    if (tid == 0) {
        int total = 0;
        for (int bt = 0; bt < m.n_band_tiles; ++bt) {
            int lo = m.plan[2 * bt], hi = m.plan[2 * bt + 1];
            total += hi > lo ? (hi - lo + 3) / 4 : 0;
        }
        int share = (total + MEL_WAVES - 1) / MEL_WAVES;
        if (share < 1) share = 1;
        int slot = 0, pos = 0;
        for (int bt = 0; bt < m.n_band_tiles; ++bt) {
            int lo = m.plan[2 * bt], hi = m.plan[2 * bt + 1];
            int rem = hi > lo ? (hi - lo + 3) / 4 : 0;
            int done = 0;
            seg->tile_first[bt] = slot;
            while (rem > 0) {
                int owner = pos / share;
                int room = (owner + 1) * share - pos;
                int take = rem < room ? rem : room;
                seg->seg_tile[slot] = bt;
                seg->seg_k0[slot] = lo + 4 * done;
                seg->seg_steps[slot] = take;
                seg->seg_wave[slot] = owner;
                ++slot;
                pos += take;
                done += take;
                rem -= take;
            }
            seg->tile_count[bt] = slot - seg->tile_first[bt];
        }
        seg->nslots = slot;
    }
    for (int i = tid; i < MEL_TILE * (PSTRIDE - NBINS); i += MEL_WAVES * 64) {
        int r = i / (PSTRIDE - NBINS), c = i % (PSTRIDE - NBINS);
        ptile[r * PSTRIDE + NBINS + c] = 0.0f;
    }

    cf tw[F::NTW];
    float2 win[F::E];
    cf ptw[F::NPAIR];
    F::load_twiddles(tw, tb.w_nc, t);
    load_window_regs<F>(win, g, t);
#pragma unroll
    for (int i = 0; i < F::NPAIR; ++i) ptw[i] = tb.w_n[t + i * F::LPF];
    __syncthreads();

    const long long tiles_per_row = (g.n_frames + MEL_TILE - 1) / MEL_TILE;
    const long long total_tiles = g.rows * tiles_per_row;
    const long long chunk = (total_tiles + gridDim.x - 1) / gridDim.x;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long end = begin + chunk < total_tiles ? begin + chunk : total_tiles;
    constexpr int FRAMES_PER_SWEEP = MEL_WAVES * F::G;
    constexpr int SWEEPS = (MEL_TILE + FRAMES_PER_SWEEP - 1) / FRAMES_PER_SWEEP;

    for (long long tile = begin; tile < end; ++tile) {
        const long long row = tile / tiles_per_row;
        const long long f0 = (tile - row * tiles_per_row) * MEL_TILE;

        // ---------------- phase A: FFT + |X|^p into the P tile
#pragma unroll 1
        for (int sw = 0; sw < SWEEPS; ++sw) {
            const int fw = (sw * MEL_WAVES + w) * F::G;     // first tile-frame of this wave
            if (fw < MEL_TILE) {
                const int fi = fw + sub;
                const long long frame = (fi < MEL_TILE) ? f0 + fi : g.n_frames;
                cf v[F::E];
                load_frame<F, true>(v, g, win, row, frame, t);
                F::run(v, lds, tw, t);
                if (fi < MEL_TILE) {
                    float* prow = ptile + fi * PSTRIDE;
#pragma unroll
                    for (int i = 0; i < F::NPAIR; ++i) {
                        const int k = t + i * F::LPF;
                        cf xa, xb;
                        F::r2c_pair(lds, k, ptw[i], xa, xb);
                        xa.x *= g.scale; xa.y *= g.scale; xb.x *= g.scale; xb.y *= g.scale;
                        prow[k] = cpow_mag(xa, m.power);
                        prow[NC - k] = cpow_mag(xb, m.power);
                    }
                    if (t == 0) {
                        cf xa, xb;
                        F::r2c_pair(lds, NC / 2, make_float2(0.0f, -1.0f), xa, xb);
                        xa.x *= g.scale; xa.y *= g.scale;
                        prow[NC / 2] = cpow_mag(xa, m.power);
                    }
                }
                wave_lds_fence();
            }
        }
        __syncthreads();

        // ---------------- phase B: block-sparse P·fb on the matrix cores
        {
            const int nslots = seg->nslots;
            const int fr = lane & 15, kq = lane >> 4;
            for (int s = 0; s < nslots; ++s) {
                if (seg->seg_wave[s] != w) continue;           // wave-uniform
                const int bt = seg->seg_tile[s];
                const int k0 = seg->seg_k0[s];
                const int steps = seg->seg_steps[s];
                const int band = bt * 16 + fr;
                const bool band_ok = band < m.n_mels;
                const float* arow = ptile + fr * PSTRIDE + k0 + kq;
                const float* bcol = m.fb + (long long)(k0 + kq) * m.n_mels + band;
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int i = 0; i < steps; ++i) {
                    const int k = k0 + kq + 4 * i;
                    float a = arow[4 * i];
                    float b = (band_ok && k < NBINS) ? bcol[(long long)4 * i * m.n_mels] : 0.0f;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                }
                float* pp = partial + s * 256 + (kq * 4) * 16 + fr;   // D[frame = kq*4+r][band = fr]
                pp[0] = acc[0]; pp[16] = acc[1]; pp[32] = acc[2]; pp[48] = acc[3];
            }
        }
        __syncthreads();

        // ---------------- phase C: reduce partials, dB, store
        {
            const int per_tile = MEL_TILE * m.n_mels;
            for (int idx = tid; idx < per_tile; idx += MEL_WAVES * 64) {
                const int fi = idx / m.n_mels;
                const int band = idx - fi * m.n_mels;
                const int bt = band >> 4;
                const int first = seg->tile_first[bt], cnt = seg->tile_count[bt];
                float sum = 0.0f;
                for (int s = 0; s < cnt; ++s) sum += partial[(first + s) * 256 + fi * 16 + (band & 15)];
                if (m.db) sum = amp_to_db(sum, m.amin, m.log10_ref);
                const long long frame = f0 + fi;
                if (frame < g.n_frames) m.out[(row * g.n_frames + frame) * m.n_mels + band] = sum;
            }
        }
        // no barrier needed here: the next tile's phase A touches scratch/ptile only, and the barrier
        // after phase A orders these partial reads before the next phase-B writes.
    }
}

template <int NC, int E>
static size_t mel_lds_bytes(int n_band_tiles) {
    using F = WaveFft<NC, E>;
    return (size_t)MEL_WAVES * F::G * F::PADDED * sizeof(cf) + (size_t)MEL_TILE * mel_pstride<NC>() * sizeof(float) +
           (size_t)(n_band_tiles + MEL_WAVES) * 256 * sizeof(float) + sizeof(SegTable);
}

template <int NC, int E>
static int launch_mel(const FrameGeom& g, const Tables& tb, MelArgs m, hipStream_t stream) {
    const size_t lds_bytes = mel_lds_bytes<NC, E>(m.n_band_tiles);
    if (lds_bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    m.pstride = mel_pstride<NC>();
    const long long tiles = g.rows * ((g.n_frames + MEL_TILE - 1) / MEL_TILE);
    int per_cu = (int)(160 * 1024 / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    long long max_blocks = (long long)device_cu_count() * per_cu;
    long long blocks = tiles < max_blocks ? tiles : max_blocks;
    if (blocks < 1) blocks = 1;
    auto kern = melspec_kernel<NC, E>;
    static bool attr_set = false;
    if (!attr_set) {
        TAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(MEL_WAVES * 64), lds_bytes, stream, g, tb, m);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// ---------------------------------------------------------------- filterbank plan
__global__ void __launch_bounds__(256) fb_plan_kernel(const float* __restrict__ fb, int n_freqs, int n_mels,
                                                      int* __restrict__ plan) {
    __shared__ int s_lo, s_hi;
    const int bt = blockIdx.x;
    if (threadIdx.x == 0) { s_lo = n_freqs; s_hi = 0; }
    __syncthreads();
    int lo = n_freqs, hi = 0;
    const int b0 = bt * 16;
    const int nb = (n_mels - b0) < 16 ? (n_mels - b0) : 16;
    for (int f = threadIdx.x; f < n_freqs; f += blockDim.x) {
        bool nz = false;
        for (int j = 0; j < nb; ++j) nz |= (fb[(long long)f * n_mels + b0 + j] != 0.0f);
        if (nz) { lo = f < lo ? f : lo; hi = (f + 1) > hi ? (f + 1) : hi; }
    }
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
    __syncthreads();
    if (threadIdx.x == 0) {
        plan[2 * bt] = s_hi > s_lo ? s_lo : 0;
        plan[2 * bt + 1] = s_hi > s_lo ? s_hi : 0;
    }
}

}  // namespace tac

extern "C" {

int tac_filterbank_plan(const float* fb, int32_t n_freqs, int32_t n_mels, int32_t* plan, void* stream) {
    using namespace tac;
    if (!fb || !plan || n_freqs <= 0 || n_mels <= 0) return TAC_E_INVALID;
    const int nt = (n_mels + 15) / 16;
    hipLaunchKernelGGL(fb_plan_kernel, dim3(nt), dim3(256), 0, (hipStream_t)stream, fb, n_freqs, n_mels, plan);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

int tac_melspec_f32(const float* wave, const float* window, const tac_stft_desc* d, float power, const float* fb,
                    const int32_t* fb_plan, int32_t n_mels, int db, float db_ref, float db_amin, float* out,
                    void* stream) {
    using namespace tac;
    if (!out || !fb || !fb_plan || n_mels <= 0) return TAC_E_INVALID;
    if (!d || !d->onesided) return d ? TAC_E_UNSUPPORTED : TAC_E_INVALID;
    FrameGeom g;
    int64_t T = 0;
    int rc = make_geometry(wave, window, d, &g, &T);
    if (rc != TAC_OK) return rc;
    if (d->n_fft > 2048) return TAC_E_UNSUPPORTED;
    const int nt = (n_mels + 15) / 16;
    if (nt > MEL_MAX_BAND_TILES) return TAC_E_UNSUPPORTED;
    Tables tb;
    rc = get_tables(d->n_fft, &tb);
    if (rc != TAC_OK) return rc;
    MelArgs m{fb, fb_plan, n_mels, nt, 0, power, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f, out};
    hipStream_t s = (hipStream_t)stream;
    switch (d->n_fft) {
        case 32: return launch_mel<16, 16>(g, tb, m, s);
        case 64: return launch_mel<32, 16>(g, tb, m, s);
        case 128: return launch_mel<64, 16>(g, tb, m, s);
        case 256: return launch_mel<128, 16>(g, tb, m, s);
        case 512: return launch_mel<256, 16>(g, tb, m, s);
        case 1024: return launch_mel<512, 16>(g, tb, m, s);
        case 2048: return launch_mel<1024, 16>(g, tb, m, s);
        default: return TAC_E_UNSUPPORTED;
    }
}

}  // extern "C"
