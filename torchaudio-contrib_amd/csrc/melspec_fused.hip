// melspec_fused.hip — STFT -> |.|^p -> mel filterbank (fp32 MFMA) -> dB in ONE kernel.
//
// Fuses the whole Sequential(*Melspectrogram(...), AmplitudeToDb()) chain of the reference
// (layers.py:307-381; functional.py:99-107, 126-128, 183-184, 291-296) so that neither the complex
// STFT (657 MB at cfg-2), the power spectrogram (328 MB) nor the linear mel tensor ever touch HBM:
// algorithmic traffic is 4*hop bytes in + 4*M bytes out per frame.
//
// Per workgroup (one CU): loop over tiles of 16 consecutive frames of one row.  LDS holds 16 frame
// buffers; a buffer is first the FFT exchange area of its frame and then, IN PLACE, the frame's
// |X|^p row — there is no separate power tile, which is what lets 16 frames fit in 160 KB.
//   phase A  every wave FFTs its frames (two in flight per wave at N=2048, fft_core.hpp) and overwrites
//            each buffer with the power row
//   phase B  P[16 x F] · fb[F x M] on v_mfma_f32_16x16x4_f32.  The triangular filters make fb
//            block-sparse: for every 16-band tile only bins [klo, khi) carry weight (the plan), so a
//            tile's K loop covers just that range.  All K-steps of all tiles are cut into equal
//            contiguous shares, one per wave, so the wide high-frequency tiles do not serialise on one
//            SIMD.  A wave's share never changes, so its B fragments (the filter weights) are loaded
//            ONCE into registers at kernel start: steady-state phase B is one ds_read + one MFMA per
//            step, no global traffic.  Shares write partial 16x16 tiles to LDS slots in a fixed order
//   phase C  fixed-order sum of a tile's partials (deterministic), optional dB epilogue, coalesced
//            512-byte row stores of out[row][frame][0..M)
#include "host_common.hpp"

#ifndef TAC_MEL_ABL
#define TAC_MEL_ABL 0    // ablation builds only: 1 = skip phase A math, 2 = skip phase B, 3 = skip phase C stores
#endif

namespace tac {

constexpr int MEL_TILE = 16;                 // frames per tile = MFMA M dimension
constexpr int MEL_MAX_BAND_TILES = 32;       // n_mels <= 512
constexpr int MEL_MAX_WAVES = 8;
constexpr int MEL_MAXS = 48;                 // K-steps per wave held in registers

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MelPlan {                             // host copy of tac_filterbank_plan's output, by value
    int lo[MEL_MAX_BAND_TILES];
    int hi[MEL_MAX_BAND_TILES];
};

struct MelArgs {
    const float* fb;       // [F][M]
    int n_mels;
    int n_band_tiles;
    float power;
    int db;
    float amin;
    float log10_ref;
    float* out;            // [rows][T][M]
};

enum { STEP_VALID = 1 << 21, STEP_FLUSH = 1 << 20 };

struct MelTables {
    int meta[MEL_MAX_WAVES][MEL_MAXS];       // k | tile << 12 | flush | valid, per wave per step
    int wave_slot0[MEL_MAX_WAVES];
    int tile_first[MEL_MAX_BAND_TILES];
    int tile_count[MEL_MAX_BAND_TILES];
};

template <int NC, int E>
struct MelCfg {
    using F = WaveFft<NC, E>;
    static constexpr int NF = (F::G == 1) ? 2 : 1;                     // frames in flight per wave
    static constexpr int FPW = NF * F::G;                              // frames per wave per tile
    static constexpr int WAVES = (MEL_TILE / FPW) >= 4 ? ((MEL_TILE / FPW) > 8 ? 8 : (MEL_TILE / FPW)) : 4;
    static constexpr int NBUF = (WAVES * FPW) > MEL_TILE ? (WAVES * FPW) : MEL_TILE;
    static constexpr int PROW = 2 * F::PADDED;                         // floats between consecutive P rows
    static_assert(PROW >= NC + 1 + 3, "P row must hold F bins + K-step overrun");
};

__host__ __device__ inline int mel_total_steps(const MelPlan& p, int n_band_tiles) {
    int total = 0;
    for (int bt = 0; bt < n_band_tiles; ++bt) total += p.hi[bt] > p.lo[bt] ? (p.hi[bt] - p.lo[bt] + 3) / 4 : 0;
    return total;
}

template <int NC, int E, bool POW2>
__global__ void __launch_bounds__((MelCfg<NC, E>::WAVES * 64))
melspec_kernel(FrameGeom g, Tables tb, MelArgs m, MelPlan plan) {
    using C = MelCfg<NC, E>;
    using F = typename C::F;
    constexpr int NF = C::NF, WAVES = C::WAVES, NBINS = NC + 1, PROW = C::PROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* bufs = reinterpret_cast<cf*>(smem_raw);                                   // NBUF frame buffers
    float* partial = reinterpret_cast<float*>(bufs + C::NBUF * F::PADDED);        // (ntiles + WAVES) x 256
    MelTables* tab = reinterpret_cast<MelTables*>(partial + (m.n_band_tiles + WAVES) * 256);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / F::LPF;
    const int t = lane % F::LPF;
    const int fr = lane & 15, kq = lane >> 4;

    // ---- one-off setup: cut the K-steps into per-wave shares
    for (int i = tid; i < MEL_MAX_WAVES * MEL_MAXS; i += WAVES * 64) (&tab->meta[0][0])[i] = 0;
    if (tid < MEL_MAX_WAVES) tab->wave_slot0[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        const int total = mel_total_steps(plan, m.n_band_tiles);
        int share = (total + WAVES - 1) / WAVES;
        if (share < 1) share = 1;
        int slot = 0, pos = 0;
        for (int bt = 0; bt < m.n_band_tiles; ++bt) {
            const int lo = plan.lo[bt];
            int rem = plan.hi[bt] > lo ? (plan.hi[bt] - lo + 3) / 4 : 0;
            int done = 0;
            tab->tile_first[bt] = slot;
            while (rem > 0) {
                const int owner = pos / share;
                const int li = pos - owner * share;
                const int room = share - li;
                const int take = rem < room ? rem : room;
                if (li == 0) tab->wave_slot0[owner] = slot;
                for (int s2 = 0; s2 < take; ++s2)
                    tab->meta[owner][li + s2] = (lo + 4 * (done + s2)) | (bt << 12) | STEP_VALID |
                                                (s2 == take - 1 ? STEP_FLUSH : 0);
                ++slot;
                pos += take;
                done += take;
                rem -= take;
            }
            tab->tile_count[bt] = slot - tab->tile_first[bt];
        }
    }
    __syncthreads();

    // ---- this wave's filter weights -> registers, once
    float breg[MEL_MAXS];
#pragma unroll
    for (int i = 0; i < MEL_MAXS; ++i) {
        const int mt = tab->meta[w][i];
        const int k = (mt & 0xfff) + kq;
        const int band = ((mt >> 12) & 0xff) * 16 + fr;
        breg[i] = ((mt & STEP_VALID) && band < m.n_mels && k < NBINS) ? m.fb[(long long)k * m.n_mels + band] : 0.0f;
    }

    cf tw[F::NTW];
    cf ptw[F::NPAIR];
    F::load_twiddles(tw, tb.w_nc, t);
#pragma unroll
    for (int i = 0; i < F::NPAIR; ++i) ptw[i] = tb.w_n[t + i * F::LPF];

    const int tiles_per_row = (int)((g.n_frames + MEL_TILE - 1) / MEL_TILE);
    const int total_tiles = (int)g.rows * tiles_per_row;
    const int chunk = (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total_tiles ? begin + chunk : total_tiles;

    cf* lds[NF];
    int fi[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        fi[f] = (w * NF + f) * F::G + sub;                  // frame index within the tile owned by this lane group
        lds[f] = bufs + fi[f] * F::PADDED;
    }
    const bool wave_has_frames = (w * C::FPW) < MEL_TILE;

    for (int tile = begin; tile < end; ++tile) {
        const int row = tile / tiles_per_row;
        const long long f0 = (long long)(tile - row * tiles_per_row) * MEL_TILE;

        // ---------------- phase A: FFT, then overwrite each frame buffer with its |X|^p row
        if (wave_has_frames) {
            cf v[NF][E];
            int tl = t;
            asm volatile("" : "+v"(tl));          // launder: window loads stay inside the loop (register budget)
            float2 win[F::E];
            load_window_regs<F>(win, g, tl);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const long long frame = (fi[f] < MEL_TILE) ? f0 + fi[f] : g.n_frames;
                load_frame<F, true>(v[f], g, win, lds[f], row, frame, t);
            }
#if TAC_MEL_ABL != 1
            F::template run<NF>(v, lds, tw, t);
#endif
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                // gather every Z this lane needs BEFORE anything is overwritten (in-place row)
#pragma unroll
                for (int i = 0; i < F::NPAIR; ++i) {
                    const int k = t + i * F::LPF;
                    v[f][2 * i] = lds[f][lds_pad(k)];
                    v[f][2 * i + 1] = lds[f][lds_pad((NC - k) & (NC - 1))];
                }
                const cf zmid = lds[f][lds_pad(NC / 2)];
                wave_lds_fence();
                float* prow = reinterpret_cast<float*>(lds[f]);
#pragma unroll
                for (int i = 0; i < F::NPAIR; ++i) {
                    const int k = t + i * F::LPF;
                    cf xa, xb;
                    F::r2c_split(v[f][2 * i], v[f][2 * i + 1], ptw[i], xa, xb);
                    xa.x *= g.scale; xa.y *= g.scale; xb.x *= g.scale; xb.y *= g.scale;
                    const float pa = xa.x * xa.x + xa.y * xa.y, pb = xb.x * xb.x + xb.y * xb.y;
                    prow[k] = POW2 ? pa : sqrtf(pa);
                    prow[NC - k] = POW2 ? pb : sqrtf(pb);
                }
                if (t == 0) {
                    const cf xm = make_float2(zmid.x * g.scale, -zmid.y * g.scale);    // X[NC/2] = conj(Z[NC/2])
                    const float pm = xm.x * xm.x + xm.y * xm.y;
                    prow[NC / 2] = POW2 ? pm : sqrtf(pm);
                }
                for (int c = t; c < 3; c += F::LPF) prow[NBINS + c] = 0.0f;           // K-step overrun columns
            }
        }
        __syncthreads();

        // ---------------- phase B: block-sparse P·fb on the matrix cores, weights from registers
#if TAC_MEL_ABL != 2
        {
            const float* abase = reinterpret_cast<const float*>(bufs) + fr * PROW + kq;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            int slot = tab->wave_slot0[w];
#pragma unroll
            for (int i = 0; i < MEL_MAXS; ++i) {
                const int mt = __builtin_amdgcn_readfirstlane(tab->meta[w][i]);
                if (mt & STEP_VALID) {
                    const float a = abase[mt & 0xfff];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[i], acc, 0, 0, 0);
                    if (mt & STEP_FLUSH) {
                        float* pp = partial + slot * 256 + (kq * 4) * 16 + fr;   // D[frame = kq*4+r][band = fr]
                        pp[0] = acc[0]; pp[16] = acc[1]; pp[32] = acc[2]; pp[48] = acc[3];
                        ++slot;
                        acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                }
            }
        }
#endif
        __syncthreads();

        // ---------------- phase C: reduce partials, dB, store
        {
            const int per_tile = MEL_TILE * m.n_mels;
            for (int idx = tid; idx < per_tile; idx += WAVES * 64) {
                const int fo = idx / m.n_mels;
                const int band = idx - fo * m.n_mels;
                const int bt = band >> 4;
                const int first = tab->tile_first[bt], cnt = tab->tile_count[bt];
                float sum = 0.0f;
                for (int s2 = 0; s2 < cnt; ++s2) sum += partial[(first + s2) * 256 + fo * 16 + (band & 15)];
                if (m.db) sum = amp_to_db(sum, m.amin, m.log10_ref);
                const long long frame = f0 + fo;
#if TAC_MEL_ABL == 3
                if (frame < g.n_frames && sum == 12345.678f) m.out[(row * g.n_frames + frame) * m.n_mels + band] = sum;
#else
                if (frame < g.n_frames) m.out[(row * g.n_frames + frame) * m.n_mels + band] = sum;
#endif
            }
        }
        // no barrier needed here: the next tile's phase A touches the frame buffers only (all phase-B reads
        // of them are behind the barrier above), and the barrier after phase A orders these partial reads
        // before the next phase-B writes.
    }
}

template <int NC, int E>
static size_t mel_lds_bytes(int n_band_tiles) {
    using C = MelCfg<NC, E>;
    return (size_t)C::NBUF * C::F::PADDED * sizeof(cf) + (size_t)(n_band_tiles + C::WAVES) * 256 * sizeof(float) +
           sizeof(MelTables);
}

template <int NC, int E>
static int launch_mel(const FrameGeom& g, const Tables& tb, MelArgs m, const MelPlan& plan, hipStream_t stream,
                      bool query_only) {
    using C = MelCfg<NC, E>;
    const size_t lds_bytes = mel_lds_bytes<NC, E>(m.n_band_tiles);
    if (lds_bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    if (m.power != 2.0f && m.power != 1.0f) return TAC_E_UNSUPPORTED;      // |X|^p, p not in {1, 2}: chain (2)+(4)
    if (mel_total_steps(plan, m.n_band_tiles) > C::WAVES * MEL_MAXS) return TAC_E_UNSUPPORTED;   // dense bank: chain (2)+(4)
    for (int bt = 0; bt < m.n_band_tiles; ++bt)
        if (plan.hi[bt] > 4000 || plan.lo[bt] < 0) return TAC_E_INVALID;
    if (query_only) return TAC_OK;
    const long long tiles = g.rows * ((g.n_frames + MEL_TILE - 1) / MEL_TILE);
    if (tiles >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    int per_cu = (int)(160 * 1024 / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    long long max_blocks = (long long)device_cu_count() * per_cu;
    long long blocks = tiles < max_blocks ? tiles : max_blocks;
    if (blocks < 1) blocks = 1;
    const bool pow2 = (m.power == 2.0f);
    auto kern = pow2 ? melspec_kernel<NC, E, true> : melspec_kernel<NC, E, false>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[pow2]) {
        TAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[pow2] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::WAVES * 64), lds_bytes, stream, g, tb, m, plan);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

static int dispatch_mel(int n_fft, const FrameGeom& g, const Tables& tb, const MelArgs& m, const MelPlan& plan,
                        hipStream_t s, bool query_only) {
    switch (n_fft) {
        case 32: return launch_mel<16, 16>(g, tb, m, plan, s, query_only);
        case 64: return launch_mel<32, 16>(g, tb, m, plan, s, query_only);
        case 128: return launch_mel<64, 16>(g, tb, m, plan, s, query_only);
        case 256: return launch_mel<128, 16>(g, tb, m, plan, s, query_only);
        case 512: return launch_mel<256, 16>(g, tb, m, plan, s, query_only);
        case 1024: return launch_mel<512, 16>(g, tb, m, plan, s, query_only);
        case 2048: return launch_mel<1024, 16>(g, tb, m, plan, s, query_only);
        default: return TAC_E_UNSUPPORTED;
    }
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

int tac_filterbank_plan(const float* fb, int32_t n_freqs, int32_t n_mels, int32_t* plan, int32_t* plan_host,
                        void* stream) {
    using namespace tac;
    if (!fb || !plan || n_freqs <= 0 || n_mels <= 0) return TAC_E_INVALID;
    const int nt = (n_mels + 15) / 16;
    hipLaunchKernelGGL(fb_plan_kernel, dim3(nt), dim3(256), 0, (hipStream_t)stream, fb, n_freqs, n_mels, plan);
    TAC_HIP(hipGetLastError());
    if (plan_host) {
        TAC_HIP(hipMemcpyAsync(plan_host, plan, sizeof(int32_t) * 2 * nt, hipMemcpyDeviceToHost, (hipStream_t)stream));
        TAC_HIP(hipStreamSynchronize((hipStream_t)stream));
    }
    return TAC_OK;
}

static int melspec_common(const float* wave, const float* window, const tac_stft_desc* d, float power, const float* fb,
                          const int32_t* fb_plan_host, int32_t n_mels, int db, float db_ref, float db_amin, float* out,
                          void* stream, bool query_only) {
    using namespace tac;
    if (!fb_plan_host || n_mels <= 0 || !d) return TAC_E_INVALID;
    if (!query_only && (!out || !fb)) return TAC_E_INVALID;
    if (!d->onesided || d->n_fft > 2048) return TAC_E_UNSUPPORTED;
    const int nt = (n_mels + 15) / 16;
    if (nt > MEL_MAX_BAND_TILES) return TAC_E_UNSUPPORTED;
    MelPlan plan;
    for (int bt = 0; bt < MEL_MAX_BAND_TILES; ++bt) {
        plan.lo[bt] = bt < nt ? fb_plan_host[2 * bt] : 0;
        plan.hi[bt] = bt < nt ? fb_plan_host[2 * bt + 1] : 0;
    }
    FrameGeom g{};
    Tables tb{};
    if (!query_only) {
        int64_t T = 0;
        int rc = make_geometry(wave, window, d, &g, &T);
        if (rc != TAC_OK) return rc;
        rc = get_tables(d->n_fft, &tb);
        if (rc != TAC_OK) return rc;
    } else if (!is_pow2(d->n_fft) || d->n_fft < 32) {
        return TAC_E_UNSUPPORTED;
    }
    MelArgs m{fb, n_mels, nt, power, db ? 1 : 0, db_amin, db ? log10f(db_ref) : 0.0f, out};
    return dispatch_mel(d->n_fft, g, tb, m, plan, (hipStream_t)stream, query_only);
}

int tac_melspec_f32(const float* wave, const float* window, const tac_stft_desc* d, float power, const float* fb,
                    const int32_t* fb_plan_host, int32_t n_mels, int db, float db_ref, float db_amin, float* out,
                    void* stream) {
    return melspec_common(wave, window, d, power, fb, fb_plan_host, n_mels, db, db_ref, db_amin, out, stream, false);
}

int tac_melspec_supported(const tac_stft_desc* d, float power, const int32_t* fb_plan_host, int32_t n_mels) {
    return melspec_common(nullptr, nullptr, d, power, nullptr, fb_plan_host, n_mels, 0, 1.0f, 1e-7f, nullptr, nullptr, true);
}

}  // extern "C"
