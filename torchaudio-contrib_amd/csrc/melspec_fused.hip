// melspec_fused.hip — STFT -> |.|^p -> mel filterbank (fp32 MFMA) -> dB in ONE kernel.
//
// Fuses the whole Sequential(*Melspectrogram(...), AmplitudeToDb()) chain of the reference
// (layers.py:307-381; functional.py:99-107, 126-128, 183-184, 291-296) so that neither the complex
// STFT (657 MB at cfg-2), the power spectrogram (328 MB) nor the linear mel tensor ever touch HBM:
// algorithmic traffic is 4*hop bytes in + 4*M bytes out per frame.
//
// A workgroup loops over tiles of TILE consecutive frames of one row.  LDS holds TILE frame buffers; a
// buffer is first the FFT exchange area of its frame and then, IN PLACE, the frame's |X|^p row — there
// is no separate power tile.
//   phase A  every wave FFTs its frames (fft_core.hpp) and overwrites each buffer with the power row
//   phase B  P[TILE x F] · fb[F x M] on v_mfma_f32_16x16x4_f32.  The triangular filters make fb
//            block-sparse: for every 16-band tile only bins [klo, khi) carry weight (the plan), so a
//            tile's K loop covers just that range.  All K-steps of all tiles are cut into equal
//            contiguous shares, one per wave, so the wide high-frequency tiles do not serialise on one
//            SIMD.  A wave's share never changes, so its B fragments (the filter weights) are loaded ONCE
//            into registers at kernel start; the P-row LDS reads of a 48-step chunk are issued together,
//            then the MFMA chain runs.  Shares write partial tiles to LDS slots in a fixed order
//   phase C  fixed-order sum of a tile's partials (deterministic), optional dB epilogue, coalesced
//            row stores of out[row][frame][0..M)
// TILE = 16 frames with one 8-wave workgroup per CU (155 KB LDS at N = 2048).  (TILE = 8 with 4-wave workgroups at 78 KB,
// so that TWO workgroups share a CU and one group's MFMA / reduction / store phases run under the other's FFT VALU
// work, measured slower at cfg-2 — 0.52 ms vs 0.37 ms: per-tile costs double while half of every MFMA's 16 rows is wasted.)
#include "mel_common.hpp"

namespace tac {

constexpr int MEL_MAX_BAND_TILES = 32;       // n_mels <= 512

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

enum { STEP_EDGE = 1 << 20 };                // step: k | tile << 12 | edge

#define TAC_STAMP(i) do {} while (0)

template <int WAVES, int CAP>
struct MelTables {
    int step[WAVES][CAP];                    // k | tile << 12 | edge  (0 when unused)
    int nsteps[WAVES];
    unsigned flush[WAVES][3];                // bit i: step i closes a partial tile (store + reset accumulator)
    int wave_slot0[WAVES];
    int tile_first[MEL_MAX_BAND_TILES];
    int tile_count[MEL_MAX_BAND_TILES];
};

__host__ __device__ inline int mel_total_steps(const MelPlan& p, int n_band_tiles) {
    int total = 0;
    for (int bt = 0; bt < n_band_tiles; ++bt) total += p.hi[bt] > p.lo[bt] ? (p.hi[bt] - p.lo[bt] + 3) / 4 : 0;
    return total;
}

template <int NC, int E, int TILE, int BUDGET, bool POW2>
__global__ void __launch_bounds__((MelCfg<NC, E, TILE, BUDGET>::WAVES * 64), 2)
melspec_kernel(FrameGeom g, Tables tb, MelArgs m, MelPlan plan) {
    using C = MelCfg<NC, E, TILE, BUDGET>;
    using F = typename C::F;
    using Tab = MelTables<C::WAVES, C::CAP>;
    constexpr int WAVES = C::WAVES, NBINS = NC + 1, PROW = C::PROW, CAP = C::CAP, SLOT = C::SLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* bufs = reinterpret_cast<cf*>(smem_raw);                                   // NBUF frame buffers
    float* partial = reinterpret_cast<float*>(bufs + C::NBUF * F::PADDED);        // (ntiles + WAVES) slots
    Tab* tab = reinterpret_cast<Tab*>(partial + (m.n_band_tiles + WAVES) * SLOT);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / F::LPF;
    const int t = lane % F::LPF;
    const int fr = lane & 15, kq = lane >> 4;

    // ---- one-off setup: cut the K-steps into per-wave shares
    for (int i = tid; i < WAVES * CAP; i += WAVES * 64) (&tab->step[0][0])[i] = 0;
    if (tid < WAVES) {
        tab->wave_slot0[tid] = 0;
        tab->nsteps[tid] = 0;
        tab->flush[tid][0] = tab->flush[tid][1] = tab->flush[tid][2] = 0;
    }
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
                for (int s2 = 0; s2 < take; ++s2) {
                    const int k = lo + 4 * (done + s2);
                    const bool edge = (k + 3 >= NBINS) || (bt * 16 + 15 >= m.n_mels);
                    tab->step[owner][li + s2] = k | (bt << 12) | (edge ? STEP_EDGE : 0);
                }
                const int last = li + take - 1;
                tab->flush[owner][last >> 5] |= 1u << (last & 31);
                tab->nsteps[owner] = li + take;
                ++slot;
                pos += take;
                done += take;
                rem -= take;
            }
            tab->tile_count[bt] = slot - tab->tile_first[bt];
        }
    }
    __syncthreads();

    // the second-dispatched half of the workgroup loses VALU arbitration to the older half on every SIMD
    // (MI355X_MICROARCH.md "Two waves per SIMD"); with one workgroup per CU a static priority bump for it
    // evens out phase A.  With two workgroups per CU the co-resident wave belongs to the OTHER group.
    if (TILE == 16 && w >= WAVES / 2) __builtin_amdgcn_s_setprio(1);

    // ---- per-wave constants of phase B (wave-uniform -> SGPRs)
    const int nsteps = __builtin_amdgcn_readfirstlane(tab->nsteps[w]);
    const unsigned flush0 = __builtin_amdgcn_readfirstlane(tab->flush[w][0]);
    const unsigned flush1 = __builtin_amdgcn_readfirstlane(tab->flush[w][1]);
    const unsigned flush2 = __builtin_amdgcn_readfirstlane(tab->flush[w][2]);
    const int slot0 = __builtin_amdgcn_readfirstlane(tab->wave_slot0[w]);

    MelFftConsts<F> fftk;
    fftk.load(tb, g, t);

    const int tiles_per_row = (int)((g.n_frames + TILE - 1) / TILE);
    const int total_tiles = (int)g.rows * tiles_per_row;
    const int chunk = (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total_tiles ? begin + chunk : total_tiles;

    // phase C walks idx = tid, tid + T, ... over (band, 4-frame group) = (idx % M, idx / M) without dividing per element
    const int c_fg0 = tid / m.n_mels, c_band0 = tid % m.n_mels;
    const int c_dfg = (WAVES * 64) / m.n_mels, c_dband = (WAVES * 64) % m.n_mels;
    // fb offset of a step for this lane = (k * M + tile * 16) (wave-uniform) + (kq * M + fr) (lane constant)
    const float* fbl = m.fb + (kq * m.n_mels + fr);
    const float* abase = reinterpret_cast<const float*>(bufs) + (fr & (TILE - 1)) * PROW + kq;

    // lane i of `step_lane[c]` keeps the (wave-uniform) word of step c*64 + i: phase B pulls each step's bin
    // offset out with v_readlane instead of a dependent LDS table read, so the P-row reads issue back to back
    constexpr int NSL = (CAP + 63) / 64;
    int step_lane[NSL];
#pragma unroll
    for (int c = 0; c < NSL; ++c) step_lane[c] = (c * 64 + lane < CAP) ? tab->step[w][c * 64 + lane] : 0;

    // this wave's filter weights for phase B (its share never changes): loaded ONCE, register-resident for the
    // kernel's lifetime — steady-state phase B touches no global memory (measured 0.38 -> 0.32 ms at cfg-2)
    float breg[CAP];
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
        const int st = tab->step[w][i];
        const int u = (st & 0xfff) * m.n_mels + ((st >> 12) & 0xff) * 16;
        breg[i] = (i < nsteps) ? fbl[u] : 0.0f;
        if (__builtin_amdgcn_readfirstlane(st) & STEP_EDGE) {                    // rare: tile/bin edge
            const bool ok = ((st & 0xfff) + kq < NBINS) && (((st >> 12) & 0xff) * 16 + fr < m.n_mels);
            breg[i] = ok ? fbl[ok ? u : 0] : 0.0f;
        }
    }
    for (int tile = begin; tile < end; ++tile) {
        const int row = tile / tiles_per_row;
        const long long f0 = (long long)(tile - row * tiles_per_row) * TILE;
        TAC_STAMP(0);

        // ---------------- phase A: FFT, then overwrite each frame buffer with its |X|^p row (mel_common.hpp)
        mel_phase_a<C, POW2>(g, bufs, fftk, w, sub, t, row, f0);
        TAC_STAMP(1);
        __syncthreads();
        TAC_STAMP(2);

        // ---------------- phase B: block-sparse P·fb on the matrix cores.  Steps past the share's end carry
        // zero weights (and read P[.][0]), so the chain needs no validity branch — only the flush points branch.
        {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            int slot = slot0;
            float a[CAP];
#pragma unroll
            for (int i = 0; i < CAP; ++i) {
                const int st = __builtin_amdgcn_readlane(step_lane[i / 64], i % 64);   // SGPR, no memory
                a[i] = abase[st & 0xfff];
            }
#pragma unroll
            for (int i = 0; i < CAP; ++i) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], breg[i], acc, 0, 0, 0);
                const unsigned word = i < 32 ? flush0 : (i < 64 ? flush1 : flush2);
                if ((word >> (i & 31)) & 1u) {                                      // wave-uniform
                    // D[frame = kq*4 + r][band = fr]; only rows < TILE are real frames
                    if (TILE == 16 || kq < 2) {
                        float* pp = partial + slot * SLOT + (kq * 4) * 16 + fr;
                        pp[0] = acc[0]; pp[16] = acc[1]; pp[32] = acc[2]; pp[48] = acc[3];
                    }
                    ++slot;
                    acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        }
        TAC_STAMP(3);
        __syncthreads();
        TAC_STAMP(4);

        // ---------------- phase C: reduce partials, dB, store.  One (band, 4-frame group) per thread so the
        // four partial sums are independent chains and every store instruction writes whole band runs.
        {
            int fg = c_fg0, band = c_band0;
            for (int idx = tid; idx < (TILE / 4) * m.n_mels; idx += WAVES * 64) {
                const int bt = band >> 4;
                const int first = tab->tile_first[bt], cnt = tab->tile_count[bt];
                float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int s2 = 0; s2 < cnt; ++s2) {
                    const float* pp = partial + (first + s2) * SLOT + (fg * 4) * 16 + (band & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[r] += pp[r * 16];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = sum[r];
                    if (m.db) v = amp_to_db(v, m.amin, m.log10_ref);
                    const long long frame = f0 + fg * 4 + r;
                    if (frame < g.n_frames) m.out[(row * g.n_frames + frame) * m.n_mels + band] = v;
                }
                band += c_dband;
                fg += c_dfg;
                if (band >= m.n_mels) { band -= m.n_mels; ++fg; }
            }
        }
        TAC_STAMP(5);
        // no barrier needed here: the next tile's phase A touches the frame buffers only (all phase-B reads
        // of them are behind the barrier above), and the barrier after phase A orders these partial reads
        // before the next phase-B writes.
    }
}

template <int NC, int E, int TILE, int BUDGET>
static size_t mel_lds_bytes(int n_band_tiles) {
    using C = MelCfg<NC, E, TILE, BUDGET>;
    return (size_t)C::NBUF * C::F::PADDED * sizeof(cf) + (size_t)(n_band_tiles + C::WAVES) * C::SLOT * sizeof(float) +
           sizeof(MelTables<C::WAVES, C::CAP>);
}

template <int NC, int E, int TILE, int BUDGET>
static int launch_mel_budget(const FrameGeom& g, const Tables& tb, MelArgs m, const MelPlan& plan, hipStream_t stream,
                             bool query_only) {
    using C = MelCfg<NC, E, TILE, BUDGET>;
    const size_t lds_bytes = mel_lds_bytes<NC, E, TILE, BUDGET>(m.n_band_tiles);
    if (lds_bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    if (m.power != 2.0f && m.power != 1.0f) return TAC_E_UNSUPPORTED;      // |X|^p, p not in {1, 2}: chain (2)+(4)
    {   // every wave's share (ceil(total / WAVES) steps) must fit its register-resident capacity
        const int total = mel_total_steps(plan, m.n_band_tiles);
        if ((total + C::WAVES - 1) / C::WAVES > C::CAP) return TAC_E_UNSUPPORTED;   // dense bank: chain (2)+(4)
    }
    for (int bt = 0; bt < m.n_band_tiles; ++bt)
        if (plan.hi[bt] > 4000 || plan.lo[bt] < 0) return TAC_E_INVALID;
    if (query_only) return TAC_OK;
    const long long tiles = g.rows * ((g.n_frames + TILE - 1) / TILE);
    if (tiles >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    int per_cu = (int)(160 * 1024 / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    long long max_blocks = (long long)device_cu_count() * per_cu;
    long long blocks = tiles < max_blocks ? tiles : max_blocks;
    if (blocks < 1) blocks = 1;
    const bool pow2 = (m.power == 2.0f);
    auto kern = pow2 ? melspec_kernel<NC, E, TILE, BUDGET, true> : melspec_kernel<NC, E, TILE, BUDGET, false>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::WAVES * 64), lds_bytes, stream, g, tb, m, plan);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// Register budget: the weights of a wave's share stay in VGPRs for the whole kernel, so the capacity is compiled
// in.  Typical 128-band mel banks need ~300 steps per workgroup (38 per wave): the 320-step build leaves the FFT
// phase 8 more registers than the 384-step one and is tried first.
template <int NC, int E, int TILE>
static int launch_mel(const FrameGeom& g, const Tables& tb, MelArgs m, const MelPlan& plan, hipStream_t stream,
                      bool query_only) {
    int rc = launch_mel_budget<NC, E, TILE, 320>(g, tb, m, plan, stream, query_only);
    if (rc == TAC_E_UNSUPPORTED) rc = launch_mel_budget<NC, E, TILE, MEL_STEP_BUDGET>(g, tb, m, plan, stream, query_only);
    return rc;
}

static int dispatch_mel(int n_fft, const FrameGeom& g, const Tables& tb, const MelArgs& m, const MelPlan& plan,
                        hipStream_t s, bool query_only) {
    switch (n_fft) {
        case 32: return launch_mel<16, 16, 16>(g, tb, m, plan, s, query_only);
        case 64: return launch_mel<32, 16, 16>(g, tb, m, plan, s, query_only);
        case 128: return launch_mel<64, 16, 16>(g, tb, m, plan, s, query_only);
        case 256: return launch_mel<128, 16, 16>(g, tb, m, plan, s, query_only);
        case 512: return launch_mel<256, 16, 16>(g, tb, m, plan, s, query_only);
        case 1024: return launch_mel<512, 16, 16>(g, tb, m, plan, s, query_only);
        case 2048: return launch_mel<1024, 16, 16>(g, tb, m, plan, s, query_only);
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
