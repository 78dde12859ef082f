// stft_small.hip — the software-pipelined STFT / spectrogram kernel for fft_length 512 and 1024, where a wave
// holds G = 4 or 2 frames (LPF = 16 or 32 lanes each, 16 complex elements per lane).
//
// Same recipe as stft_pipe_kernel (stft_kernels.hip): the NEXT unit's samples are requested before the current
// unit's rows are stored, every store is unconditional and nontemporal (so the wait for the samples is an exact
// vmcnt), the window comes from LDS, and the R2C split exchanges only the upper half of each spectrum.  A unit is
// G consecutive frames of one row, so its G output rows are adjacent in memory and leave as ONE contiguous run of
// 16-byte stores; a unit with frames in the padding, or past the end of its row, takes the gather path and clamps
// its run to the live rows.
#include "host_common.hpp"
#include "mel_lanes.hpp"

namespace tac {

constexpr int SM_WAVES = 8;    // one workgroup per CU; its waves draw units from a workgroup counter (see stft_pipe_kernel)
typedef float sm_f4 __attribute__((ext_vector_type(4)));

// MEL: the fused Melspectrogram (-> AmplitudeToDb) form — the unit's |X|^p rows stay in LDS (pitch sm_mel_pitch) and
// are contracted with a band-sparse filterbank there by the frame's LPF lanes (mel_lanes.hpp, S steps per band); the
// mel rows are staged behind them and stored instead of the spectrogram rows.
__host__ __device__ constexpr int sm_mel_pitch(int nc) { return (nc + 1 + 3 + 3) & ~3; }
#ifndef TAC_SM_FLY
#define TAC_SM_FLY 6
#endif
constexpr int SM_FLY = TAC_SM_FLY;       // contraction steps in flight (the packed layout depends on it: groups of slots)
constexpr int SM_MAX_STEPS_1024 = 20;    // fft_length 1024, float32 input: four-tap steps of the widest band (80 bins; LM_MAX_STEPS = 12 elsewhere)
}  // namespace tac
#include "stft_small3.hpp"
namespace tac {

template <int NC, int MODE, bool MEL, int S>
__global__ void __launch_bounds__(SM_WAVES * 64, 2)
stft_small_kernel(FrameGeom g, Tables tb, StftEpilogue ep, LaneMel mel) {
    constexpr int E = 16;
    using F = WaveFft<NC, E>;
    constexpr int LPF = F::LPF, G = F::G;
    static_assert(G >= 2 && radix_at(NC, 0) == E, "several frames per wave, one first-pass butterfly per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / LPF, t = lane % LPF;
    constexpr int WAVE_SLOTS = ((G * F::PADDED + 1) / 2) * 2;
    cf* const wbase = smem + w * WAVE_SLOTS;
    cf* const lds = wbase + sub * F::PADDED;
    // window pairs, one 144-byte row per first-pass column (8 conflict-free ds_read_b128 per lane)
    constexpr int WROW = E + 2;
    cf* const wlds = smem + SM_WAVES * WAVE_SLOTS;
    for (int m = threadIdx.x; m < NC; m += SM_WAVES * 64) wlds[(m % LPF) * WROW + (m / LPF)] = window_pair(g, m);

    cf tw[F::NTW];
    cf ptw[F::NPAIR];
    F::load_twiddles(tw, tb.w_nc, t);
#pragma unroll
    for (int i = 0; i < F::NPAIR; ++i) ptw[i] = tb.w_n[t + i * LPF];

    const int T = (int)g.n_frames;
    const int upr = (T + G - 1) / G;                       // units per row
    const int total = (int)g.rows * upr;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total ? begin + chunk : total;
    constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);
    constexpr int NST = (((G * LENF) >> 2) + 63) / 64;    // 16-byte wave-stores per unit
    const float hscale = 0.5f * g.scale;                  // the R2C split returns 2·X

    cf raw[E];
    // every lane group requests its own frame; the unit takes the fast path only if ALL its frames are interior
    auto prefetch = [&](int unit) -> bool {
        const int urow = unit / upr;
        const int frame = (unit - urow * upr) * G + sub;
        const long long start = (long long)frame * g.hop - g.center_pad;
        const bool ok = g.vec2_ok && frame < T && start >= 0 && start + F::N <= g.length;
        const bool all_ok = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        if (all_ok) {
            const cf* src = reinterpret_cast<const cf*>(g.wave + (long long)urow * g.row_stride + start);
#pragma unroll
            for (int q = 0; q < E; ++q) raw[q] = src[t + q * LPF];
        }
        return all_ok;
    };
    unsigned* const next_unit = reinterpret_cast<unsigned*>(wlds + LPF * WROW);
    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + SM_WAVES);
    constexpr int PITCH = sm_mel_pitch(NC), MEL_OFF = G * PITCH + 8;
    static_assert(!MEL || MEL_OFF + 4 + G * LM_MAX_MELS <= 2 * WAVE_SLOTS, "mel rows fit the wave's area");
    int* const mlo = reinterpret_cast<int*>(next_unit + 4);                // MEL: first bins [slot][lane]; the weights
    float* const mwl = reinterpret_cast<float*>(mlo + lm_desc_ints(LPF));
    if constexpr (MEL) lane_mel_load_tables<S, LPF, SM_FLY>(mlo, mwl, mel, threadIdx.x, SM_WAVES * 64);
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    bool pre = false;
    int unit = begin + w;
    if (unit < end) pre = prefetch(unit);
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the loop is entered with nothing in flight
    __syncthreads();

    NoStamp st;
    while (unit < end) {
        const int nxt = grab();
        const int urow = unit / upr;
        const int uframe0 = (unit - urow * upr) * G;
        cf v[1][E];
        cf* const ldsv[1] = {lds};
        if (pre) {
            const sm_f4* wp = reinterpret_cast<const sm_f4*>(wlds + t * WROW);
            sm_f4 wv[E / 2];
#pragma unroll
            for (int i = 0; i < E / 2; ++i) wv[i] = wp[i];
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                v[0][2 * i] = cmul_elem(raw[2 * i], mkc(wv[i].x, wv[i].y));
                v[0][2 * i + 1] = cmul_elem(raw[2 * i + 1], mkc(wv[i].z, wv[i].w));
            }
        } else {
            load_frame<F, false>(v[0], g, nullptr, lds, urow, uframe0 + sub, t);    // padding / past-the-end frames
        }
        F::template run<1, NoStamp, true>(v, ldsv, tw, t, st, t);                    // lower-half spectrum stays in registers

        __builtin_amdgcn_sched_barrier(0);
        {
            pre = false;
            if (nxt < end) pre = prefetch(nxt);
        }
        __builtin_amdgcn_sched_barrier(0);

        const long long g0 = ((long long)urow * T + uframe0) * (MEL ? mel.n_mels : LENF);
        const int a = MEL ? 0 : (int)(g0 & 3);
        float* const stage = reinterpret_cast<float*>(wbase) + a;        // LDS and global share their 16-byte phase
        float* const srow = stage + sub * (MEL ? PITCH : LENF);
        {
            cf xa[F::NPAIR], xb[F::NPAIR], xm, unused;
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) {
                const int k = t + i * LPF;
                const cf zk = v[0][F::reg_of_spectrum(i)];
                const cf zm = (i == 0) ? F::r2c_partner(lds, k, zk) : lds[lds_pad(NC - k)];
                if constexpr (MODE != 0) {                                // xa[i] = (|X[k]|^2, |X[NC-k]|^2), no spectra formed
                    xa[i] = cscale(F::r2c_power_x2(zk, zm, ptw[i]), hscale * hscale);
                } else {
                    F::r2c_split_x2(zk, zm, ptw[i], xa[i], xb[i]);
                    xa[i] = cscale(xa[i], hscale); xb[i] = cscale(xb[i], hscale);
                }
            }
            F::r2c_pair(lds, NC / 2, mkc(0.0f, -1.0f), xm, unused);
            xm = cscale(xm, hscale);
            wave_lds_fence();                                             // every Z of this unit is in registers
#pragma unroll
            for (int i = 0; i < F::NPAIR; ++i) {
                const int k = t + i * LPF;
                if constexpr (MODE == 0) {
                    reinterpret_cast<cf*>(srow)[k] = xa[i];
                    reinterpret_cast<cf*>(srow)[NC - k] = xb[i];
                } else {
                    srow[k] = spectral_row_value<MODE>(xa[i].x, ep);
                    srow[NC - k] = spectral_row_value<MODE>(xa[i].y, ep);
                }
            }
            if (t == 0) {
                if constexpr (MODE == 0) reinterpret_cast<cf*>(srow)[NC / 2] = xm;
                else srow[NC / 2] = spectral_row_value<MODE>(cnorm2(xm), ep);
            }
            wave_lds_fence();
        }
        const int nlive = (T - uframe0) < G ? (T - uframe0) : G;
        if constexpr (MEL) {
            const int am = (int)(g0 & 3);
            float* const mstage = reinterpret_cast<float*>(wbase) + MEL_OFF + am;
            lane_mel_contract<S, LPF, SM_FLY>(srow, NC + 1, mlo, mwl, t, mel, mstage + sub * mel.n_mels);
            wave_lds_fence();
            lane_mel_store<(G * LM_MAX_MELS + 255) / 256>(mstage, am, nlive * mel.n_mels, mel.out + g0, lane);
            wave_lds_fence();   // next iteration's first-pass writes must follow these reads
            unit = nxt;
            continue;
        }
        // the unit's live rows leave as 1 + NST + 1 unconditional nontemporal stores (lanes past the end repeat a neighbour)
        const int len = nlive * LENF;
        float* const gdst = ep.out + g0;
        const int npre = (4 - a) & 3;
        const int nchunks = (len - npre) >> 2;
        {
            const int hmax = (npre > 1 ? npre : 1) - 1;
            const int hi = lane < hmax ? lane : hmax;
            gdst[hi] = stage[hi];
        }
        const sm_f4* const s4 = reinterpret_cast<const sm_f4*>(stage + npre);
        sm_f4* const g4 = reinterpret_cast<sm_f4*>(gdst + npre);
        const int last = nchunks - 1;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = (lane + 64 * i) < last ? (lane + 64 * i) : last;
            __builtin_nontemporal_store(s4[c], g4 + c);
        }
        {
            const int r = len - npre - 4 * nchunks;
            const int rmax = (r > 1 ? r : 1) - 1;
            const int ti = len - 1 - (lane < rmax ? lane : rmax);
            gdst[ti] = stage[ti];
        }
        wave_lds_fence();   // next iteration's first-pass writes must follow these reads
        unit = nxt;
    }
}

// waves per CU of the stft_small3 form: 0 = the two-wave kernel (TAC_SMALL2=1), else TAC_SM3_WAVES or 16
static int small3_waves() {
    static const int wv = [] {
        const char* two = getenv("TAC_SMALL2");
        if (two && two[0] == '1') return 0;
        const char* e = getenv("TAC_SM3_WAVES");
        return e ? atoi(e) : 16;
    }();
    return wv;
}

template <int NC, int MODE>
static int launch_small(const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, hipStream_t stream) {
    using F = WaveFft<NC, 16>;
    const long long units = g.rows * ((g.n_frames + F::G - 1) / F::G);
    if (units >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    constexpr int WAVE_SLOTS = ((F::G * F::PADDED + 1) / 2) * 2;
    const size_t bytes = (size_t)SM_WAVES * WAVE_SLOTS * sizeof(cf) + (size_t)F::LPF * 18 * sizeof(cf) + 16;
    long long blocks = (units + SM_WAVES - 1) / SM_WAVES;
    const long long cap = (long long)device_cu_count();
    if (blocks > cap) blocks = cap;
    if (small3_waves()) {
        // three / four waves per SIMD (stft_small3.hpp); TAC_SMALL2=1 keeps the two-wave kernel below
        auto go = [&](auto k3, int W) -> int {
            const size_t b3 = small3_lds_bytes<NC>(W);
            long long bl = (units + W - 1) / W;
            if (bl > cap) bl = cap;
            TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(k3), (int)b3));
            hipLaunchKernelGGL(k3, dim3((unsigned)bl), dim3(W * 64), b3, stream, g, tb, ep, LaneMel{}, (const void*)nullptr,
                               (const float*)nullptr);
            TAC_HIP(hipGetLastError());
            return TAC_OK;
        };
        // complex rows are bound by their stores (12 waves measure like the two-wave kernel, 16 slower); real rows gain 8-10 %
        // from the fourth wave per SIMD (profiles/r03/ab_stream3.txt)
        const int wv = getenv("TAC_SM3_WAVES") ? small3_waves() : (MODE == 0 ? 12 : 16);
        if (wv == 12) return go(stft_small3_kernel<NC, MODE, false, 1, 12>, 12);
        return go(stft_small3_kernel<NC, MODE, false, 1, 16>, 16);
    }
    if constexpr (NC < 256) {
        return TAC_E_UNSUPPORTED;                            // fft_length 256 has no two-wave form: the generic kernel
    } else {
        auto kern = stft_small_kernel<NC, MODE, false, 1>;
        if (bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SM_WAVES * 64), bytes, stream, g, tb, ep, LaneMel{});
        TAC_HIP(hipGetLastError());
        return TAC_OK;
    }
}

template <int NC>
static size_t small_lds_bytes() {
    using F = WaveFft<NC, 16>;
    constexpr int WAVE_SLOTS = ((F::G * F::PADDED + 1) / 2) * 2;
    return (size_t)SM_WAVES * WAVE_SLOTS * sizeof(cf) + (size_t)F::LPF * 18 * sizeof(cf) + 16;
}

template <int NC, int MODE, int S>
static int launch_small_mel(const FrameGeom& g, const Tables& tb, const LaneMel& mel, hipStream_t stream) {
    using F = WaveFft<NC, 16>;
    const long long units = g.rows * ((g.n_frames + F::G - 1) / F::G);
    if (units >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const size_t bytes = small_lds_bytes<NC>() + lm_lds_bytes(F::LPF, mel.wtot);
    if (bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    long long blocks = (units + SM_WAVES - 1) / SM_WAVES;
    const long long cap = (long long)device_cu_count();
    if (blocks > cap) blocks = cap;
    if (small3_waves()) {
        const size_t b3 = small3_lds_bytes<NC>(12) + lm_lds_bytes(F::LPF, mel.wtot);
        if (b3 <= 160 * 1024) {
            auto k3 = stft_small3_kernel<NC, MODE, true, S, 12>;
            long long bl = (units + 12 - 1) / 12;
            if (bl > cap) bl = cap;
            TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(k3), (int)b3));
            hipLaunchKernelGGL(k3, dim3((unsigned)bl), dim3(12 * 64), b3, stream, g, tb,
                               StftEpilogue{nullptr, 1, 1, MODE == 1 ? 2.0f : 1.0f, 0, 0.0f, 0.0f}, mel, (const void*)nullptr,
                               (const float*)nullptr);
            TAC_HIP(hipGetLastError());
            return TAC_OK;
        }
    }
    if constexpr (NC < 256) {
        return TAC_E_UNSUPPORTED;
    } else {
        auto kern = stft_small_kernel<NC, MODE, true, S>;
        if (bytes > 64 * 1024) TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SM_WAVES * 64), bytes, stream, g, tb,
                           StftEpilogue{nullptr, 1, 1, MODE == 1 ? 2.0f : 1.0f, 0, 0.0f, 0.0f}, mel);
        TAC_HIP(hipGetLastError());
        return TAC_OK;
    }
}

// int16 PCM / mu-law codes read by the fused kernel itself (power 2 only: one instantiation per format and band length)
template <int NC, int S, int FMT>
static int launch_small_mel_coded(FrameGeom g, const Tables& tb, const LaneMel& mel, hipStream_t stream, const void* samples,
                                  const float* lut) {
    using F = WaveFft<NC, 16>;
    const long long units = g.rows * ((g.n_frames + F::G - 1) / F::G);
    if (units >= 0x7fffffffLL || !small3_waves()) return TAC_E_UNSUPPORTED;
    const size_t b3 = small3_lds_bytes<NC>(12) + lm_lds_bytes(F::LPF, mel.wtot) + 1024;
    if (b3 > 160 * 1024) return TAC_E_UNSUPPORTED;
    {                                                                      // sample pairs fetched as one access of the format
        const uintptr_t pair = FMT == FMT_I16 ? 4 : (FMT == FMT_MULAW_U8 ? 2 : 8);
        g.vec2_ok = ((g.hop & 1) == 0) && ((g.center_pad & 1) == 0) && ((g.row_stride & 1) == 0) &&
                    ((reinterpret_cast<uintptr_t>(samples) & (pair - 1)) == 0);
    }
    auto k3 = stft_small3_kernel<NC, 1, true, S, 12, FMT>;
    long long bl = (units + 12 - 1) / 12;
    if (bl > device_cu_count()) bl = device_cu_count();
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(k3), (int)b3));
    hipLaunchKernelGGL(k3, dim3((unsigned)bl), dim3(12 * 64), b3, stream, g, tb, StftEpilogue{nullptr, 1, 1, 2.0f, 0, 0.0f, 0.0f}, mel,
                       samples, lut);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

template <int NC, int FMT>
static int launch_small_mel_coded_nc(const FrameGeom& g, const Tables& tb, const LaneMel& mel, int S, hipStream_t stream,
                                     const void* samples, const float* lut) {
    switch (S) {
#define TAC_SM_CASE(SS) case SS: return launch_small_mel_coded<NC, SS, FMT>(g, tb, mel, stream, samples, lut);
        TAC_SM_CASE(2) TAC_SM_CASE(4) TAC_SM_CASE(6) TAC_SM_CASE(8) TAC_SM_CASE(10) TAC_SM_CASE(12)
#undef TAC_SM_CASE
        default: return S > LM_MAX_STEPS && S <= SM_MAX_STEPS_1024 ? TAC_E_UNSUPPORTED : TAC_E_INVALID;   // (wide-band tables: float32 input only)
    }
}

template <int FMT>
static int launch_small_mel_coded_fmt(int n_fft, const FrameGeom& g, const Tables& tb, const LaneMel& mel, int S, hipStream_t stream,
                                      const void* samples, const float* lut) {
    if (n_fft == 256) return launch_small_mel_coded_nc<128, FMT>(g, tb, mel, S, stream, samples, lut);
    if (n_fft == 512) return launch_small_mel_coded_nc<256, FMT>(g, tb, mel, S, stream, samples, lut);
    return launch_small_mel_coded_nc<512, FMT>(g, tb, mel, S, stream, samples, lut);
}

template <int NC>
static int launch_small_mel_nc(const FrameGeom& g, const Tables& tb, float power, const LaneMel& mel, int S, hipStream_t stream) {
    const bool p2 = power == 2.0f;
    switch (S) {                                                           // steps per band (pack_small: even values)
#define TAC_SM_CASE(SS) case SS: return p2 ? launch_small_mel<NC, 1, SS>(g, tb, mel, stream) : launch_small_mel<NC, 2, SS>(g, tb, mel, stream);
        TAC_SM_CASE(2) TAC_SM_CASE(4) TAC_SM_CASE(6) TAC_SM_CASE(8) TAC_SM_CASE(10) TAC_SM_CASE(12)
#undef TAC_SM_CASE
        default: break;
    }
    if constexpr (NC == 512) {                                             // fft_length 1024: bands up to 80 bins (40- and 64-band banks; round 6)
        switch (S) {
#define TAC_SM_CASE(SS) case SS: return p2 ? launch_small_mel<NC, 1, SS>(g, tb, mel, stream) : launch_small_mel<NC, 2, SS>(g, tb, mel, stream);
            TAC_SM_CASE(14) TAC_SM_CASE(16) TAC_SM_CASE(18) TAC_SM_CASE(20)
#undef TAC_SM_CASE
            default: break;
        }
    }
    return TAC_E_INVALID;
}

// The fused Melspectrogram (+dB) chain for fft_length 512 / 1024 (melspec_sparse.hip's entry points call these).
int launch_small_mel_entry(int n_fft, const FrameGeom& g, const Tables& tb, float power, const float* wpack, const int* desc,
                           const int32_t* info_host, int n_mels, int db, float amin, float log10_ref, float* out,
                           hipStream_t stream, int fmt, const void* samples, const float* lut) {
    const int lanes = n_fft / 32;
    if ((n_fft != 256 && n_fft != 512 && n_fft != 1024) || !lane_mel_info_ok(info_host, lanes, SM_FLY, n_fft == 1024 ? SM_MAX_STEPS_1024 : LM_MAX_STEPS))
        return TAC_E_INVALID;
    if (n_mels < LM_MIN_MELS || n_mels > LM_MAX_MELS) return TAC_E_UNSUPPORTED;
    const LaneMel mel{wpack, desc, info_host[1], info_host[0], n_mels, db, amin, log10_ref, out, info_host[5] ? 1 : 0};
    if (fmt != FMT_F32) {
        if (power != 2.0f) return TAC_E_UNSUPPORTED;
        switch (fmt) {
            case FMT_I16: return launch_small_mel_coded_fmt<FMT_I16>(n_fft, g, tb, mel, info_host[4], stream, samples, lut);
            case FMT_MULAW_U8: return launch_small_mel_coded_fmt<FMT_MULAW_U8>(n_fft, g, tb, mel, info_host[4], stream, samples, lut);
            default: return launch_small_mel_coded_fmt<FMT_MULAW_I64>(n_fft, g, tb, mel, info_host[4], stream, samples, lut);
        }
    }
    if (n_fft == 256) return launch_small_mel_nc<128>(g, tb, power, mel, info_host[4], stream);
    return n_fft == 512 ? launch_small_mel_nc<256>(g, tb, power, mel, info_host[4], stream)
                        : launch_small_mel_nc<512>(g, tb, power, mel, info_host[4], stream);
}

int pack_small(int n_fft, const std::vector<float>& h, int n_freqs, int n_mels, float* wpack, int wpack_cap, int32_t* desc,
               int desc_cap, int32_t* info_host, hipStream_t stream, bool to_host) {
    if ((n_fft != 256 && n_fft != 512 && n_fft != 1024) || n_freqs != n_fft / 2 + 1) return TAC_E_UNSUPPORTED;
    if (n_fft == 256 && !small3_waves()) return TAC_E_UNSUPPORTED;        // (TAC_SMALL2=1: the three-phase kernel's layout)
    const int lanes = n_fft / 32;
    const size_t base = n_fft == 256 ? small3_lds_bytes<128>(12) : (n_fft == 512 ? small_lds_bytes<256>() : small_lds_bytes<512>());
    return pack_lane_mel(h, n_freqs, n_mels, lanes, sm_mel_pitch(n_fft / 2), 2, SM_FLY, n_fft == 1024 ? SM_MAX_STEPS_1024 : LM_MAX_STEPS, base, wpack, wpack_cap, desc, desc_cap,
                         info_host, stream, to_host);
}

template <int NC>
static int launch_small_mode(int pmode, const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, hipStream_t s) {
    switch (pmode) {
        case 0: return launch_small<NC, 0>(g, tb, ep, s);
        case 1: return launch_small<NC, 1>(g, tb, ep, s);
        case 2: return launch_small<NC, 2>(g, tb, ep, s);
        case 3: return launch_small<NC, 3>(g, tb, ep, s);
        default: return launch_small<NC, 4>(g, tb, ep, s);
    }
}

// Entry used by stft_kernels.hip's dispatcher (fft_length 512 / 1024): TAC_E_UNSUPPORTED when this form does not
// apply (two-sided output, |X|^p with p outside {1, 2}) so that the generic kernel takes over.
int try_launch_small(int n_fft, const FrameGeom& g, const Tables& tb, const StftEpilogue& ep, int mode,
                     hipStream_t stream) {
    if (!ep.onesided) return TAC_E_UNSUPPORTED;
    int pmode = -1;
    if (mode == 0) pmode = 0;
    else if (ep.power == 2.0f) pmode = ep.db ? 3 : 1;
    else if (ep.power == 1.0f) pmode = ep.db ? 4 : 2;
    if (pmode < 0) return TAC_E_UNSUPPORTED;
    if (n_fft == 1024) return launch_small_mode<512>(pmode, g, tb, ep, stream);
    if (n_fft == 512) return launch_small_mode<256>(pmode, g, tb, ep, stream);
    if (n_fft == 256) return launch_small_mode<128>(pmode, g, tb, ep, stream);
    return TAC_E_UNSUPPORTED;
}

}  // namespace tac
