// mel_lanes.hpp — the band-sparse filterbank contraction of the kernels that keep a wave's |X|^p rows in LDS in output
// order (stft_n400.hip: 8 lanes per frame; stft_small.hip: 16 / 32 lanes per frame), fused behind their R2C split:
// Melspectrogram (-> AmplitudeToDb) in one launch (reference layers.py:307-381, functional.py:172-184, :277-296).
//
// Lane l of a frame's LANES owns bands l, LANES + l, 2 LANES + l, ... (band slot i = band / LANES); every band is S
// four-tap steps (S = the longest band of the bank, a template parameter; shorter bands zero-padded, first bins rounded
// down to a multiple of four, runs that would leave the row shifted down): one 16-byte weight read, one 16-byte row
// read and two packed FMAs per step.  The contraction is latency, not work (a first version with one LDS round trip
// per step took a third of the fft_length-400 kernel): slots are processed in groups of lm_group(S, FLY) whose reads
// are all issued before the first FMA, and a group's first bins are read one group earlier.
// pack_lane_mel builds:  desc = first bin [slot][lane] (zero-padded to whole groups plus one), wpack =
// [slot][step][lane][4 taps] (zero-padded to whole groups).
// (Measured and dropped: cutting the bands into equal 8-bin chunks dealt to the lanes in order, partial sums combined
// with ds_add_f32 into the LDS row — no wasted taps, but the atomics cost more than the taps they save: 0.36 vs 0.157 ms
// at 512 / 80 bands.)
#pragma once
#include "host_common.hpp"

#include <algorithm>
#include <vector>

namespace tac {

struct LaneMel {
    const float* wpack;
    const int* desc;
    int nslot, wtot, n_mels, db;
    float amin, log10_ref;
    float* out;                 // [rows][T][n_mels]
    int rev;                    // cell LANES i + l holds band n_mels - 1 - (LANES i + l) (info[5]; pack_lane_mel)
};

constexpr int LM_MAX_MELS = 128, LM_MIN_MELS = 8, LM_MAX_STEPS = 12;
constexpr int LM_MAX_STEPS_WAVE = 36;   // LANES = 64 (one frame per wave, two slots): bands up to 144 bins — 128 mel bands over 2049 bins
constexpr int LM_MARK = 1000;                                              // info_host[2] = LM_MARK + lanes per frame
typedef float lm_f4 __attribute__((ext_vector_type(4)));

// slots per group: about FLY steps (8 registers each) in flight
__host__ __device__ constexpr int lm_group(int S, int FLY) { return FLY / S < 1 ? 1 : (FLY / S > 8 ? 8 : FLY / S); }
// slots the weights cover (whole groups) and slots the first-bin table covers (one group more: a group's first bins
// are read a group ahead)
__host__ __device__ constexpr int lm_weight_slots(int nslot, int S, int FLY) {
    return ((nslot + lm_group(S, FLY) - 1) / lm_group(S, FLY)) * lm_group(S, FLY);
}
__host__ __device__ constexpr int lm_padded_slots(int nslot, int S, int FLY) { return lm_weight_slots(nslot, S, FLY) + lm_group(S, FLY); }
// first bins [padded slot][lane] followed by the padded slots' own step counts (round 3: a slot stops at ITS longest band)
__host__ __device__ constexpr int lm_desc_ints(int lanes) { return lanes * (LM_MAX_MELS / lanes + 8) + 32; }
// LDS of the fused form behind a kernel's own: first bins + packed weights
inline size_t lm_lds_bytes(int lanes, int wtot) {
    return (size_t)lm_desc_ints(lanes) * sizeof(int) + (((size_t)wtot + 3) & ~(size_t)3) * sizeof(float);
}

// tables into LDS (all threads of the workgroup; the caller's barrier follows)
template <int S, int LANES, int FLY>
__device__ __forceinline__ void lane_mel_load_tables(int* mlo, float* mwl, const LaneMel& mel, int tid, int nthreads) {
    for (int i = tid; i < (LANES + 1) * lm_padded_slots(mel.nslot, S, FLY); i += nthreads) mlo[i] = mel.desc[i];
    for (int i = tid; i < mel.wtot; i += nthreads) mwl[i] = mel.wpack[i];
}

// N steps of one slot (N <= S), FLY in flight at a time
template <int N, int LANES, int FLY>
__device__ __forceinline__ void lm_slot_steps(const lm_f4* wp, const lm_f4* pp, cf& acc0, cf& acc1) {
#pragma unroll
    for (int c0 = 0; c0 < N; c0 += FLY) {
        lm_f4 wv[FLY], pv[FLY];
#pragma unroll
        for (int j = 0; j < FLY; ++j)
            if (c0 + j < N) {
                wv[j] = wp[(c0 + j) * LANES];
                pv[j] = pp[c0 + j];
            }
#pragma unroll
        for (int j = 0; j < FLY; ++j)
            if (c0 + j < N) {
                acc0 = __builtin_elementwise_fma(mkc(wv[j].x, wv[j].y), mkc(pv[j].x, pv[j].y), acc0);
                acc1 = __builtin_elementwise_fma(mkc(wv[j].z, wv[j].w), mkc(pv[j].z, pv[j].w), acc1);
            }
        asm volatile("" : "+v"(acc0), "+v"(acc1) : : "memory");              // the next chunk's reads stay behind these FMAs
    }
}
// wave-uniform `pairs` (1 .. P) -> the body unrolled for min(2 pairs, S) steps
template <int P, int S, int LANES, int FLY>
__device__ __forceinline__ void lm_slot_dispatch(int pairs, const lm_f4* wp, const lm_f4* pp, cf& acc0, cf& acc1) {
    if constexpr (P <= 1) {
        lm_slot_steps<(2 < S ? 2 : S), LANES, FLY>(wp, pp, acc0, acc1);
    } else {
        if (pairs >= P) lm_slot_steps<(2 * P < S ? 2 * P : S), LANES, FLY>(wp, pp, acc0, acc1);
        else lm_slot_dispatch<P - 1, S, LANES, FLY>(pairs, wp, pp, acc0, acc1);
    }
}

// One frame's row (srow: 16-byte aligned, `bins` values, at least three floats of slack behind them) -> its mel (dB) row
template <int S, int LANES, int FLY>
__device__ __forceinline__ void lane_mel_contract(float* srow, int bins, const int* mlo, const float* mwl, int l,
                                                  const LaneMel& mel, float* mrow) {
    constexpr int GS = lm_group(S, FLY);
    const bool fast_db = mel.amin >= 1.1754944e-38f;                       // (uniform) hardware log2 unless the clamp admits denormals
    const float ten_log10_ref = 10.0f * mel.log10_ref;
    if (l < 3) srow[bins + l] = 0.0f;                                      // slack taps carry zero weights: keep them finite
    wave_lds_fence();
    if constexpr (S > FLY || GS == 1) {
        // One slot after the other, FLY of its steps in flight at a time.  A slot runs ITS OWN number of steps (the longest band
        // among its lanes, in pairs; wave-uniform, from the table behind the first bins) through a switch over fully unrolled
        // bodies: the zero-padded tail of the uniform-S layout — 40 % of the steps of an 80-band bank at fft_length 512 — is
        // neither read nor multiplied.
        const int* const msteps = mlo + LANES * lm_padded_slots(mel.nslot, S, FLY);
        int lo_c = mlo[l];
#pragma unroll 1
        for (int i0 = 0; i0 < mel.nslot; ++i0) {
            cf acc0 = mkc(0.0f, 0.0f), acc1 = mkc(0.0f, 0.0f);
            const lm_f4* wp = reinterpret_cast<const lm_f4*>(mwl) + i0 * (S * LANES) + l;
            const lm_f4* pp = reinterpret_cast<const lm_f4*>(srow + lo_c);
            lo_c = mlo[LANES * (i0 + 1) + l];                                // (table padded by one group)
            const int pairs = __builtin_amdgcn_readfirstlane(msteps[i0]);    // steps of this slot / 2, 1 .. (S + 1) / 2
            lm_slot_dispatch<(S + 1) / 2, S, LANES, FLY>(pairs, wp, pp, acc0, acc1);
            float val = (acc0.x + acc0.y) + (acc1.x + acc1.y);
            if (mel.db) val = fast_db ? amp_to_db_fast(val, mel.amin, ten_log10_ref) : amp_to_db(val, mel.amin, mel.log10_ref);
            const int cell = LANES * i0 + l;
            if (cell < mel.n_mels) mrow[mel.rev ? mel.n_mels - 1 - cell : cell] = val;
        }
        return;
    }
    int lo_g[GS];
#pragma unroll
    for (int q = 0; q < GS; ++q) lo_g[q] = mlo[LANES * q + l];
#pragma unroll 1
    for (int i0 = 0; i0 < mel.nslot; i0 += GS) {
        lm_f4 wv[GS][S], pv[GS][S];
        const lm_f4* wp = reinterpret_cast<const lm_f4*>(mwl) + i0 * (S * LANES) + l;
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            const lm_f4* pp = reinterpret_cast<const lm_f4*>(srow + lo_g[q]);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                wv[q][j] = wp[(q * S + j) * LANES];
                pv[q][j] = pp[j];
            }
        }
#pragma unroll
        for (int q = 0; q < GS; ++q) lo_g[q] = mlo[LANES * (i0 + GS + q) + l];         // the next group's (table padded by one group)
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            cf acc0 = mkc(0.0f, 0.0f), acc1 = mkc(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                acc0 = __builtin_elementwise_fma(mkc(wv[q][j].x, wv[q][j].y), mkc(pv[q][j].x, pv[q][j].y), acc0);
                acc1 = __builtin_elementwise_fma(mkc(wv[q][j].z, wv[q][j].w), mkc(pv[q][j].z, pv[q][j].w), acc1);
            }
            float val = (acc0.x + acc0.y) + (acc1.x + acc1.y);
            if (mel.db) val = fast_db ? amp_to_db_fast(val, mel.amin, ten_log10_ref) : amp_to_db(val, mel.amin, mel.log10_ref);
            const int cell = LANES * (i0 + q) + l;
            if (cell < mel.n_mels) mrow[mel.rev ? mel.n_mels - 1 - cell : cell] = val;
        }
    }
}

// The unit's `len` staged floats (mstage shares the 16-byte phase of gdst) leave as 1 + NSTM + 1 unconditional
// nontemporal stores — a FIXED number (surplus ones repeat the last chunk), so that the wait for the next unit's
// samples, requested before them, stays a counted vmcnt.  len >= 8.
template <int NSTM>
__device__ __forceinline__ void lane_mel_store(const float* mstage, int am, int len, float* gdst, int lane) {
    const int npre = (4 - am) & 3;
    const int nchunks = (len - npre) >> 2;
    {
        const int hmax = (npre > 1 ? npre : 1) - 1;
        const int hi = lane < hmax ? lane : hmax;
        gdst[hi] = mstage[hi];
    }
    const lm_f4* const s4 = reinterpret_cast<const lm_f4*>(mstage + npre);
    lm_f4* const g4 = reinterpret_cast<lm_f4*>(gdst + npre);
    const int last = nchunks - 1;
#pragma unroll
    for (int i = 0; i < NSTM; ++i) {
        const int c = (lane + 64 * i) < last ? (lane + 64 * i) : last;
        __builtin_nontemporal_store(s4[c], g4 + c);
    }
    {
        const int r = len - npre - 4 * nchunks;
        const int rmax = (r > 1 ? r : 1) - 1;
        const int ti = len - 1 - (lane < rmax ? lane : rmax);
        gdst[ti] = mstage[ti];
    }
}

// Host: the layout above for a (n_freqs x n_mels) bank `h`, `lanes` lanes per frame, rows of `pitch` floats, S a
// multiple of `step_quantum`, groups sized for `fly` steps in flight (the kernel's FLY).
// TAC_E_UNSUPPORTED when the bank does not fit (fewer than 8 / more than 128 bands, bands wider than 4 LM_MAX_STEPS).
inline int pack_lane_mel(const std::vector<float>& h, int n_freqs, int n_mels, int lanes, int pitch, int step_quantum, int fly,
                         int max_steps, size_t base_lds, float* wpack, int wpack_cap, int32_t* desc, int desc_cap, int32_t* info_host,
                         hipStream_t stream, bool to_host = false) {
    if (n_mels < LM_MIN_MELS || n_mels > LM_MAX_MELS) return TAC_E_UNSUPPORTED;
    const int nslot = (n_mels + lanes - 1) / lanes;
    std::vector<int> blo(n_mels, 0), bhi(n_mels, 0);                       // per band
    int S = 1;
    for (int m = 0; m < n_mels; ++m) {
        int l0 = n_freqs, h0 = 0;
        for (int f = 0; f < n_freqs; ++f)
            if (h[(size_t)f * n_mels + m] != 0.0f) { l0 = f < l0 ? f : l0; h0 = f + 1; }
        if (h0 > l0) {
            blo[m] = l0;
            bhi[m] = h0;
            S = std::max(S, (h0 - (l0 & ~3) + 3) / 4);
        }
    }
    // Round 6: a bank whose band count is not a multiple of `lanes` and whose bands widen with their number (every mel bank) is laid out
    // from its widest end — cell c = lanes i + l holds band n_mels - 1 - c (info[5] = 1, LaneMel::rev) — so that the widest bands share
    // slot 0 instead of defining a partly empty last slot; every slot runs the steps of ITS longest band (80 bands on 32 lanes: 14
    // steps per frame instead of 17)
    const bool rev = (n_mels % lanes) != 0 && (bhi[n_mels - 1] - blo[n_mels - 1]) > (bhi[0] - blo[0]);
    auto band_of = [&](int c) { return rev ? n_mels - 1 - c : c; };
    std::vector<int> lo(nslot * lanes, 0), hi(nslot * lanes, 0);           // per cell
    for (int c = 0; c < n_mels; ++c) {
        lo[c] = blo[band_of(c)];
        hi[c] = bhi[band_of(c)];
    }
    S = ((S + step_quantum - 1) / step_quantum) * step_quantum;            // (the kernels are instantiated for these values of S only)
    if (S > max_steps || 4 * S > pitch) return TAC_E_UNSUPPORTED;          // bands too wide: the unfused chain
    const int pslot = lm_padded_slots(nslot, S, fly);
    if ((lanes + 1) * pslot > desc_cap || (lanes + 1) * pslot > lm_desc_ints(lanes)) return TAC_E_UNSUPPORTED;
    const long long wtot = 4LL * lanes * S * lm_weight_slots(nslot, S, fly);
    if (wtot > wpack_cap || base_lds + lm_lds_bytes(lanes, (int)wtot) > 160 * 1024) return TAC_E_UNSUPPORTED;
    std::vector<float> wp((size_t)wtot, 0.0f);
    std::vector<int32_t> dd((size_t)(lanes + 1) * pslot, 0);               // first bins, then the slots' step counts in pairs
    for (int i = 0; i < pslot; ++i) dd[(size_t)lanes * pslot + i] = 1;
    for (int i = 0; i < nslot; ++i)
        for (int l = 0; l < lanes; ++l) {
            const int m = lanes * i + l;
            int first = lo[m] & ~3;
            if (first + 4 * S > pitch) first = pitch - 4 * S;              // keep the padded run inside the row
            for (int j = 0; j < S; ++j)
                for (int u = 0; u < 4; ++u) {
                    const int bin = first + 4 * j + u;
                    const bool live = m < n_mels && bin >= lo[m] && bin < hi[m];
                    wp[(((size_t)i * S + j) * lanes + l) * 4 + u] = live ? h[(size_t)bin * n_mels + band_of(m)] : 0.0f;
                }
            dd[lanes * i + l] = first;
            if (m < n_mels && hi[m] > lo[m]) {
                const int need = (hi[m] - first + 3) / 4;                  // steps this band takes from its (possibly shifted) first bin
                int32_t& pr = dd[(size_t)lanes * pslot + i];
                pr = std::max<int32_t>(pr, std::min((need + 1) / 2, (S + 1) / 2));
            }
        }
    if (to_host) {                                                         // (tac_melbank_pack_host: host buffers, no device)
        std::copy(wp.begin(), wp.end(), wpack);
        std::copy(dd.begin(), dd.end(), desc);
    } else {
        TAC_HIP(hipMemcpyAsync(wpack, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        TAC_HIP(hipMemcpyAsync(desc, dd.data(), dd.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        TAC_HIP(hipStreamSynchronize(stream));
    }
    info_host[0] = (int32_t)wtot;
    info_host[1] = nslot;
    info_host[2] = LM_MARK + lanes;
    info_host[3] = S * nslot;
    info_host[4] = S;
    info_host[5] = rev ? 1 : 0;
    for (int i = 6; i < 8; ++i) info_host[i] = 0;
    return TAC_OK;
}

// what a launcher checks before trusting a pack
inline bool lane_mel_info_ok(const int32_t* info, int lanes, int fly, int max_steps = LM_MAX_STEPS) {
    return info[2] == LM_MARK + lanes && info[1] >= 1 && info[1] <= LM_MAX_MELS / lanes + (LM_MAX_MELS % lanes ? 1 : 0) &&
           info[4] >= 1 && info[4] <= max_steps && info[0] == 4 * lanes * info[4] * lm_weight_slots(info[1], info[4], fly);
}

}  // namespace tac
