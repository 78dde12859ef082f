// melspec_stream3.hpp — the fused fft_length-2048 chain (STFT -> |X|^p -> band-sparse mel filterbank -> dB, one launch; the
// benchmark kernel since round 3) with THREE waves per SIMD.  Replaces reference layers.py:307-381 / functional.py:36-38,
// 58-72, 172-184, 291-296 for this size.
//
// melspec_stream_kernel (round 2) runs two 239-register waves per SIMD, each rotating two frames; its counters say that both
// the VALU (64 %) and the LDS (44 %) idle while the SIMD's two waves are stuck at the same time.  This form trades the
// software rotation for a third hardware wave: twelve waves per workgroup (<= 168 VGPRs), ONE frame per wave, the
// |X|^2 row written in place over the frame's exchange area (which is what lets twelve waves fit the LDS:
// 12 x 8.7 KB + 20 KB of weights), no instruction of one frame interleaved with another's.
#pragma once
#include <type_traits>

#include "melspec_stream.hpp"

namespace tac {

constexpr int S3_WAVES = 12;
constexpr int S3_TW2L_BYTES = 64 * 12 * 8;     // pass-2 twiddle table of the FAST1 kernels (behind the window table)

// Frozen in round 5 (the knobs, their ablation branches and the piece-layout / 15-wave / stamp variants of this kernel are in
// tools/ablation/stream3_lab_knobs_r05.patch with the measurement that decided each): first exchange XOR-swizzled instead of
// padded and partner exchange dense (no bank conflicts on either side: complex rows -1.05 %, power rows -1.3 %, fused kernel
// -0.2 %); exchange read-backs as single ds_read_b64 (2 LDS cycles each; hipcc merges pairs into ds_read2_b64 at 8: -1.2 ... -2.0 %);
// the |X|^2 row written as eight ds_write2st64_b32 (-0.65 %); the lane's eight R2C twiddles in registers (-0.6 %, -1.5 % together
// with the row pairs); the wave's first frame requested behind the table loads; frames dealt from the middle of the chunk.
// bytes of one wave's exchange area: the transform needs NC dense slots (swizzled layout), the gather path of edge frames still
// writes padded slots (<= 8696 B), and areas are 128-byte multiples so that the swizzle is one XOR on the byte address
template <class F>
__host__ __device__ constexpr int s3_xa_bytes() {
    return ((F::NC + F::NC / 16 - 1) * (int)sizeof(cf) + 127) & ~127;
}

// exchange areas + bank weights + pass-1 twiddles + frame counter + R2C twiddle table + window table [+ mu-law table]
template <int NC, int E>
__host__ __device__ inline size_t stream3_lds_bytes(int wtot, int waves, bool coded) {
    using C = StreamCfg<NC, E>;
    size_t xa = (size_t)s3_xa_bytes<typename C::F>();
    return (size_t)waves * xa + (((size_t)wtot * 4 + 15) & ~(size_t)15) + ST_TW_BYTES + 16 + 64 * (C::F::NPAIR + E) * sizeof(cf) + (coded ? 1024 : 0);
}

// LDS byte offset of a pointer into the workgroup's shared memory (the low half of its flat address)
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)reinterpret_cast<unsigned long long>(p); }
// one ds_read_b64 that hipcc's load/store optimizer cannot pair into a ds_read2_b64; the caller waits with lds_wait_all()
template <int BYTE_OFF>
__device__ __forceinline__ cf lds_read_b64_single(unsigned addr) {
    cf r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(BYTE_OFF) : "memory");
    return r;
}
// s_waitcnt lgkmcnt(0) that the listed registers' uses cannot move above
__device__ __forceinline__ void lds_wait_all(cf (&a)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                   "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                 :: "memory");
}
__device__ __forceinline__ void lds_wait_all8(cf* a) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 :: "memory");
}
__device__ __forceinline__ void lds_wait_all(cf (&a)[8], cf& b) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b)
                 :: "memory");
}

// ---- the first exchange without padding.  Lane tt's first-pass output k (element 16 tt + k) lives in slot
// 16 tt + (k ^ (tt & 15)): for a fixed k the sixteen lanes of a ds_write_b64 group hit sixteen different slots mod 16 (all 32
// banks once); lane t of pass 1 reads element t + 64 q from block B = (t >> 4) + 4 q, slot 16 B + ((t & 15) ^ (B & 15)), and
// B & 15 = (t >> 4) + 4 (q & 3): the 32 lanes of a ds_read_b64 group cover two whole 16-slot blocks 16 slots apart (all 64 banks
// once), and a lane needs just FOUR base addresses (by q & 3) + immediate offsets 512 q.  The padded layout costs one conflict
// cycle per single-b64 read-back (lanes 0 and 31 of a group share a bank) and ~2-way conflicts on the 16-byte first-pass writes.
struct S3Swz {
    unsigned w0;          // byte address of this lane's first-pass slot for k = 0; slot k is w0 ^ (8 k) (areas are 128-byte aligned)
    unsigned r[4];        // byte address of this lane's pass-1 operand q = j (j = q & 3), minus 512 q
    __device__ __forceinline__ void init(const cf* xa, int t) {
        const unsigned base = lds_offset_of(xa);
        w0 = base + 128u * (unsigned)t + 8u * (unsigned)(t & 15);
        const unsigned a = (unsigned)t >> 4, k0 = (unsigned)t & 15u;
#pragma unroll
        for (unsigned j = 0; j < 4; ++j) r[j] = base + 8u * (16u * a + (k0 ^ (a + 4u * j)));
    }
};
__device__ __forceinline__ void s3_write_pass0_swz(const cf (&v)[16], const S3Swz& z) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned addr = z.w0 ^ (8u * (unsigned)k);
        asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v[k]) : "memory");
    }
}
__device__ __forceinline__ void s3_readback_pass1_swz(cf (&v)[16], const S3Swz& z) {
    auto rd = [&](auto qc) { constexpr int q = decltype(qc)::value; v[q] = lds_read_b64_single<512 * q>(z.r[q & 3]); };
    rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
    rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
    rd(std::integral_constant<int, 8>{}); rd(std::integral_constant<int, 9>{}); rd(std::integral_constant<int, 10>{}); rd(std::integral_constant<int, 11>{});
    rd(std::integral_constant<int, 12>{}); rd(std::integral_constant<int, 13>{}); rd(std::integral_constant<int, 14>{}); rd(std::integral_constant<int, 15>{});
    lds_wait_all(v);
}
// ... and the partner exchange dense: the upper half of the spectrum at xa[o] (lanes write and read consecutive slots)
template <class F>
__device__ __forceinline__ void s3_r2c_partners_dense(const cf (&v)[16], cf* xa, cf (&zm)[8], cf& zmid, int t) {
    constexpr int NC = F::NC;
    static_assert(NC == 1024 && F::E == 16 && radix_at(NC, 2) == 4, "16 . 16 . 4 plan");
    wave_lds_fence();
#pragma unroll
    for (int b = 0; b < 4; ++b)                                               // butterfly b of the last pass: outputs k = 2, 3 (upper half)
#pragma unroll
        for (int k = 2; k < 4; ++k) xa[t + 64 * b + 256 * k] = v[b * 4 + k];
    wave_lds_fence();
    const unsigned pa = lds_offset_of(xa + (NC - t - 7 * 64));                // partner of pair 7; pair p sits (7 - p) * 64 slots above
    auto rd = [&](auto pc) { constexpr int p = decltype(pc)::value; zm[p] = lds_read_b64_single<(7 - p) * 64 * 8>(pa); };
    rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
    rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
    zmid = lds_read_b64_single<0>(lds_offset_of(xa + NC / 2));
    lds_wait_all(zm, zmid);
    if (t == 0) zm[0] = v[F::reg_of_spectrum(0)];                            // (slot NC is not stored: bin 0 pairs with itself)
}

// ---- the two exchange read-backs of the one-frame-per-wave front end, shared by melspec_stream3_kernel and stft_stream3_kernel
// (1) operands of pass 1 back from the exchange area
template <class F>
__device__ __forceinline__ void s3_readback_pass1(cf (&v)[16], const cf* xa, int t) {
    const unsigned ra = lds_offset_of(xa + lds_pad(t));
    auto rd = [&](auto qc) { constexpr int q = decltype(qc)::value; v[q] = lds_read_b64_single<lds_pad_c(q * (F::NC / 16)) * 8>(ra); };
    rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
    rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
    rd(std::integral_constant<int, 8>{}); rd(std::integral_constant<int, 9>{}); rd(std::integral_constant<int, 10>{}); rd(std::integral_constant<int, 11>{});
    rd(std::integral_constant<int, 12>{}); rd(std::integral_constant<int, 13>{}); rd(std::integral_constant<int, 14>{}); rd(std::integral_constant<int, 15>{});
    lds_wait_all(v);
}
// (2) the R2C partners Z[NC - k] of the lane's eight pairs and Z[NC / 2], from the upper half of the spectrum in the exchange area
template <class F>
__device__ __forceinline__ void s3_read_partners(const cf (&v)[16], const cf* xa, cf (&zm)[8], cf& zmid, int t) {
    constexpr int NC = F::NC;
    static_assert(F::NPAIR == 8, "eight partners per lane");
    const unsigned pa = lds_offset_of(xa + lds_pad(NC - t) - lds_pad_c(7 * F::LPF));      // partner of pair 7; pair p sits (7 - p) * 68 slots above
    auto rd = [&](auto pc) { constexpr int p = decltype(pc)::value; zm[p] = lds_read_b64_single<lds_pad_c((7 - p) * F::LPF) * 8>(pa); };
    rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
    rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
    zmid = lds_read_b64_single<0>(lds_offset_of(xa + lds_pad(NC / 2)));
    lds_wait_all(zm, zmid);
    if (t == 0) zm[0] = v[F::reg_of_spectrum(0)];
}
// ... after the butterflies of the last pass: the half write of the spectrum, then the partners
template <class F>
__device__ __forceinline__ void s3_r2c_partners(const cf (&v)[16], cf* xa, cf (&zm)[8], cf& zmid, int t) {
    s3_r2c_partners_dense<F>(v, xa, zm, zmid, t);
}
// eight / sixteen complex values 64 slots apart starting at `first` (a padded slot address of the exchange area)
template <int N0, int CNT>
__device__ __forceinline__ void s3_read_strided(cf* dst, const cf* first) {
    static_assert(CNT == 8 || CNT == 16, "eight or sixteen values");
    const unsigned ra = lds_offset_of(first);
    auto rd = [&](auto qc) { constexpr int q = decltype(qc)::value; if constexpr (q < CNT) dst[q] = lds_read_b64_single<lds_pad_c((N0 + q) * 64) * 8>(ra); };
    rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
    rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
    rd(std::integral_constant<int, 8>{}); rd(std::integral_constant<int, 9>{}); rd(std::integral_constant<int, 10>{}); rd(std::integral_constant<int, 11>{});
    rd(std::integral_constant<int, 12>{}); rd(std::integral_constant<int, 13>{}); rd(std::integral_constant<int, 14>{}); rd(std::integral_constant<int, 15>{});
    if constexpr (CNT == 16) lds_wait_all(*reinterpret_cast<cf(*)[16]>(dst));
    else lds_wait_all8(dst);
}

// ---- set-up shared by the one-frame-per-wave kernels (this file, stft_stream3.hpp): the loop-invariant tables into LDS.
// The pass-1 twiddle sets and the R2C twiddles come from the library's table cache already in their LDS layout
// (Tables::s3img, host_common.hip) and are copied as 16-byte chunks; the window pairs are read lane-contiguously (one
// 8-byte load per element, no index arithmetic for a full-length window) and land as [element >> 1][lane][element & 1]
// with the transform's scale folded in.  issue() only loads (so that every global load of the set-up is in flight before
// the first LDS store), store() only stores.  Round 3 computed each entry's table index per thread: ~700 instructions
// per wave and a 64-bit division for the chunk size, 4.3-4.7 us per launch (profiles/r03/ubench/stream3_cycles.txt).
typedef float pf4 __attribute__((ext_vector_type(4)));
template <class F, int THREADS>
struct S3Setup {
    static constexpr int TOT = S3_IMG_TW1_F4 + S3_IMG_PTW_F4;
    static constexpr int NIMG = (TOT + THREADS - 1) / THREADS;
    static constexpr int NWIN = (64 * F::E + THREADS - 1) / THREADS;
    static_assert(F::N == 2048 && F::E == 16 && F::NPAIR == 8 && S3_IMG_TW_STRIDE == ST_TW_STRIDE, "image layout of get_tables()");
    pf4 img[NIMG];
    cf win[NWIN];
    __device__ __forceinline__ void issue(const FrameGeom& g, const Tables& tb, int tid) {
#pragma unroll
        for (int u = 0; u < NIMG; ++u) {
            const int c = tid + u * THREADS;
            img[u] = reinterpret_cast<const pf4*>(tb.s3img)[c < TOT ? c : TOT - 1];
        }
        if (g.win_length == F::N) {
            const cf* w2 = reinterpret_cast<const cf*>(g.window);
#pragma unroll
            for (int u = 0; u < NWIN; ++u) {
                const int idx = tid + u * THREADS;
                win[u] = w2[idx < 64 * F::E ? idx : 0];
            }
        } else {
#pragma unroll
            for (int u = 0; u < NWIN; ++u) {
                const int idx = tid + u * THREADS;
                win[u] = window_pair(g, idx < 64 * F::E ? idx : 0);
            }
        }
    }
    // element m = lane + 64 q of the frame (complex pair of samples 2m, 2m + 1) is lane `lane`'s first-pass register q
    __device__ __forceinline__ void store(float* twlds, cf* ptwl, cf* winl, float half, int tid) const {
#pragma unroll
        for (int u = 0; u < NIMG; ++u) {
            const int c = tid + u * THREADS;
            if (c < S3_IMG_TW1_F4) reinterpret_cast<pf4*>(twlds)[c] = img[u];
            else if (c < TOT) reinterpret_cast<pf4*>(ptwl)[c - S3_IMG_TW1_F4] = img[u];
        }
#pragma unroll
        for (int u = 0; u < NWIN; ++u) {
            const int idx = tid + u * THREADS;
            const int q = idx >> 6, tt = idx & 63;
            if (idx < 64 * F::E) winl[((q >> 1) * 64 + tt) * 2 + (q & 1)] = cscale(win[u], half);
        }
    }
};

// FAST1: 0 = any bank the lane layout takes; otherwise the number of steps of slot 1 in the (4, FAST1)-step two-slot layout of
// a 128-band bank (what tac_melbank_pack produces for the standard mel banks): the contraction fully unrolled
// FMT: sample format of the frame load (FMT_*): float32, int16 PCM, mu-law codes as uint8 / int64 (converted in registers)
// WAVES: waves per workgroup (= per CU): 12, three per SIMD
template <int NC, int E, bool POW2, int FMT, int FAST1, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (WAVES + 3) / 4)
melspec_stream3_kernel(FrameGeom g, Tables tb, StreamArgs m) {
    using C = StreamCfg<NC, E>;
    using F = typename C::F;
    constexpr int NBINS = C::NBINS;
    constexpr int XA_BYTES = s3_xa_bytes<F>();
    static_assert(XA_BYTES >= (int)(C::PROW * 4), "the power row fits the exchange area");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* const xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);
    float* const prow = reinterpret_cast<float*>(xa);                               // the frame's |X|^2 row, in place
    const long long chunk = m.chunk;                     // ceil(total / gridDim.x), from the host (a 64-bit division is ~130 scalar instructions)
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < m.total ? begin + chunk : m.total;
    const int nloc = endl > begin ? (int)(endl - begin) : 0;
    const unsigned T = (unsigned)g.n_frames;
    const int t = lane;

    // the i-th frame dealt is frame (i + nloc / 2) mod nloc of the chunk: a chunk is often a whole row, whose first and last frames
    // touch the padding and take the slower gather path — they are dealt in the middle of the run, not as its tail
    const int deal_shift = nloc >> 1;
    auto place = [&](int i) { const int j = i + deal_shift; return j < nloc ? j : j - nloc; };
    cf v[E];
    int mode = 0, row = 0;
    long long fr = 0;
    auto request = [&](int i) {
        i = i < nloc ? i : nloc - 1;
        const unsigned gf = (unsigned)(begin + place(i));
        const unsigned r = gf / T;
        row = (int)r;
        fr = (long long)(gf - r * T);
        const long long start = fr * (long long)g.hop - g.center_pad;
        const bool ok = g.vec2_ok && start >= 0 && start + F::N <= g.length;
        mode = ok ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + F::N <= g.length ? cs : g.length - F::N;
        const long long off = (long long)row * g.row_stride + cs;                  // in samples
        if constexpr (FMT == FMT_F32) {
            const cf* src = reinterpret_cast<const cf*>(static_cast<const float*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = src[t + q * F::LPF];
        } else if constexpr (FMT == FMT_I16) {                              // a pair of samples = one dword
            const unsigned* src = reinterpret_cast<const unsigned*>(static_cast<const short*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[q].x = __uint_as_float(src[t + q * F::LPF]);
        } else if constexpr (FMT == FMT_MULAW_U8) {                         // a pair of codes = one 16-bit load
            const unsigned short* src = reinterpret_cast<const unsigned short*>(static_cast<const unsigned char*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) v[q].x = __uint_as_float((unsigned)src[t + q * F::LPF]);
        } else {                                                            // int64 codes: the low dword of each
            const int* src = reinterpret_cast<const int*>(static_cast<const long long*>(m.samples) + off);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                v[q].x = __int_as_float(src[4 * (t + q * F::LPF)]);
                v[q].y = __int_as_float(src[4 * (t + q * F::LPF) + 2]);
            }
        }
    };
    // ---- tables into LDS.  Every global load of the set-up is issued before the first LDS store (loads of one loop iteration
    //      used to wait for the previous iteration's: a dozen serialized L2 round trips, 4.3 us of a 110 us kernel; now ~one)
    float* const wlds = reinterpret_cast<float*>(smem_raw + (size_t)WAVES * XA_BYTES);
    float* const twlds = wlds + ((m.wtot + 3) & ~3);
    unsigned* const next_frame = reinterpret_cast<unsigned*>(twlds + ST_TW_BYTES / 4);
    cf* const ptwl = reinterpret_cast<cf*>(next_frame + 4);
    cf* const winl = ptwl + 64 * F::NPAIR;
    // (2X -> scale * X once, in the window; int16 PCM samples enter as integers: their 2^-15 goes in as well)
    const float half = 0.5f * g.scale * (FMT == FMT_I16 ? (1.0f / 32768.0f) : 1.0f);
    constexpr int WCH = 4;                                 // 16-byte weight chunks per thread and round
    const int n4 = (m.wtot + 3) >> 2;                      // (the pack is zero-padded to whole 16-byte chunks)
    pf4 wreg[WCH];
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const int c = tid + u * WAVES * 64;
        wreg[u] = reinterpret_cast<const pf4*>(m.wl)[c < n4 ? c : n4 - 1];
    }
    S3Setup<F, WAVES * 64> setup;
    setup.issue(g, tb, tid);
    cf tw2[3];
    {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw2[q] = all[twiddles_before(NC, E, 2) + q];
    }
    int lo_s[ST_MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < ST_MAX_SLOTS; ++s) lo_s[s] = s < m.nslot ? m.lo[s * 64 + lane] : 0;
    float lutv = 0.0f;
    if (FMT >= FMT_MULAW_U8) lutv = m.lut[tid & 255];
    __builtin_amdgcn_sched_barrier(0);
    // the wave's first frame is requested BEHIND the table loads (one in-order vmcnt: the tables' wait then leaves these sixteen
    // loads outstanding) and ahead of the LDS stores and the barrier: its HBM latency runs behind the rest of the set-up
    if (nloc > 0) request(w);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stores
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const int c = tid + u * WAVES * 64;
        if (c < n4) reinterpret_cast<pf4*>(wlds)[c] = wreg[u];
    }
    for (int c = tid + WCH * WAVES * 64; c < n4; c += WAVES * 64)          // banks with more than 48 KB of weights: the rest, plainly
        reinterpret_cast<pf4*>(wlds)[c] = reinterpret_cast<const pf4*>(m.wl)[c];
    if (tid == 0) *next_frame = WAVES;
    // pass-1 twiddle sets and the window pairs of its sixteen first-pass elements (scale folded in), as [read u][lane] 16-byte pairs: every
    // ds_read_b128 of the wave is one contiguous kilobyte
    setup.store(twlds, ptwl, winl, half, tid);
    cf ptw_regs[F::NPAIR];                                  // the lane's eight R2C twiddles (16 VGPRs: the kernel sits at 156 of 168)
#pragma unroll
    for (int p = 0; p < F::NPAIR; ++p) ptw_regs[p] = tb.w_n[t + p * F::LPF];
    float* const lutlds = reinterpret_cast<float*>(winl + 64 * E);                 // mu-law decode table (coded inputs)
    // FAST1 kernels (the standard 128-band banks: 18 - 20 KB of weights, float32 samples; round 6): the pass-2 twiddles with the last
    // pass's W_16 constants multiplied in, W_NC^((t + 64 b) q), b < 4, q = 1 .. 3, as [u < 6][lane] pairs in LDS instead of three hoisted
    // registers + nine constant multiplies per frame: -1.2 % same process (profiles/r06/ab/batch24_mel_pass2_table.txt).  The general
    // kernels keep the registers: their LDS belongs to the bank (tables of up to 42 steps)
    constexpr bool TW2L = FAST1 > 0;
    cf* const tw2l = reinterpret_cast<cf*>(lutlds);
    if constexpr (TW2L) {
        for (int i = tid; i < 64 * 12; i += WAVES * 64) {
            const int tt = i & 63, e = i >> 6, b = e / 3, q = e % 3 + 1;
            tw2l[((e >> 1) * 64 + tt) * 2 + (e & 1)] = tb.w_nc[((tt + 64 * b) * q) & 1023];
        }
    }
    if (FMT >= FMT_MULAW_U8 && tid < 256) lutlds[tid] = lutv;
    __syncthreads();
    if (nloc <= 0) return;

    typedef float f4 __attribute__((ext_vector_type(4)));
    const bool fast_db = m.amin >= 1.1754944e-38f;
    const float ten_log10_ref = 10.0f * m.log10_ref;
    // the requested registers as float sample pairs (still unwindowed): PCM integers / decoded codes
    auto decode = [&]() {
        if constexpr (FMT == FMT_I16) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int bits = __float_as_int(v[q].x);
                v[q] = mkc((float)(short)(bits & 0xffff), (float)(bits >> 16));
            }
        } else if constexpr (FMT == FMT_MULAW_U8) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const unsigned bits = __float_as_uint(v[q].x);
                v[q] = mkc(lutlds[bits & 0xffu], lutlds[(bits >> 8) & 0xffu]);
            }
        } else if constexpr (FMT == FMT_MULAW_I64) {
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = mkc(lutlds[__float_as_uint(v[q].x) & 0xffu], lutlds[__float_as_uint(v[q].y) & 0xffu]);
        }
    };
    struct Fetch {                                        // sample access of the gather path, in the same units as `decode`
        const void* base;
        const float* lut;
        __device__ __forceinline__ float operator()(long long row_offset, int j) const {
            if constexpr (FMT == FMT_F32) return static_cast<const float*>(base)[row_offset + j];
            else if constexpr (FMT == FMT_I16) return (float)static_cast<const short*>(base)[row_offset + j];
            else if constexpr (FMT == FMT_MULAW_U8) return lut[static_cast<const unsigned char*>(base)[row_offset + j]];
            else return lut[(unsigned)static_cast<const long long*>(base)[row_offset + j] & 0xffu];
        }
    };
    auto fma4 = [](f4 wv, f4 pv, cf& a0, cf& a1) {
        a0 = __builtin_elementwise_fma(mkc(wv.x, wv.y), mkc(pv.x, pv.y), a0);
        a1 = __builtin_elementwise_fma(mkc(wv.z, wv.w), mkc(pv.z, pv.w), a1);
    };
    S3Swz swz;
    swz.init(xa, t);
    int i = w;
    // diagnostics (tac_debug_clock_probe): shader cycles and 100 MHz ticks of wave 0's frame loop -> the clock the kernel ran at
    unsigned long long probe_c = 0, probe_w = 0;
    if (m.probe && w == 0) {
        probe_c = __builtin_readcyclecounter();
        probe_w = wall_clock64();
    }
    while (i < nloc) {
        // the next frame of this wave (the counter's answer travels with the stage's other LDS traffic)
        unsigned ask = 0;
        if (lane == 0) ask = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- s0: window, pass 0, exchange
        if (mode == 1) {
            decode();
        } else {                                            // frames touching the padding gather their (decoded) samples first
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v, g, nullptr, xa, row, fr, tz, Fetch{m.samples, lutlds});
        }
        {
            cf win[E];
            const f4* wl = reinterpret_cast<const f4*>(winl) + t;
#pragma unroll
            for (int u = 0; u < E / 2; ++u) {
                const f4 x = wl[u * 64];
                win[2 * u] = mkc(x.x, x.y);
                win[2 * u + 1] = mkc(x.z, x.w);
            }
            Dft<16>::run_windowed(v, win);
        }
        wave_lds_fence();
        cf tw1[16];
        {
            const f4* tl = reinterpret_cast<const f4*>(twlds + (t & 15) * ST_TW_STRIDE);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4 x = tl[u];
                tw1[2 * u] = mkc(x.x, x.y);
                tw1[2 * u + 1] = mkc(x.z, x.w);
            }
        }
        s3_write_pass0_swz(v, swz);
        wave_lds_fence();
        s3_readback_pass1_swz(v, swz);
        // ---- s12
        F::template pass_twiddle<1, true>(v, tw1);
        F::template pass_butterflies<1>(v);
        F::exchange_1_2_in_registers(v);
        if constexpr (TW2L) {
            const f4* tl2 = reinterpret_cast<const f4*>(tw2l) + t;
            cf w2[12];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const f4 x = tl2[u * 64];
                w2[2 * u] = mkc(x.x, x.y);
                w2[2 * u + 1] = mkc(x.z, x.w);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                cmul_x2(v[4 * b + 1], w2[3 * b], v[4 * b + 2], w2[3 * b + 1]);
                v[4 * b + 3] = cmul(v[4 * b + 3], w2[3 * b + 2]);
            }
        } else {
            F::template pass_twiddle<2, true>(v, tw2);
        }
        F::template pass_butterflies<2>(v);
        cf zm[F::NPAIR], zmid;
        s3_r2c_partners<F>(v, xa, zm, zmid, t);
        // ---- s3: R2C split -> |X|^p; the row overwrites the exchange area once every lane holds its partners
        cf pw[F::NPAIR];
        const cf (&ptw)[F::NPAIR] = ptw_regs;
#pragma unroll
        for (int p = 0; p < F::NPAIR; p += 2)
            r2c_power_pair_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], v[F::reg_of_spectrum(p + 1)], zm[p + 1], ptw[p + 1], pw[p], pw[p + 1]);
        const float pmid = 4.0f * cnorm2(zmid);
        wave_lds_fence();                                                   // all partner reads are in registers
        {
            // bins t + 64 p and t + 64 (p + 1) are 256 bytes apart: one ds_write2st64_b32 per pair of pairs (6 LDS cycles for two
            // dwords against 4 + 4), lower half ascending from &prow[t], upper half descending onto &prow[NC - t - 448]
            float lo[F::NPAIR], hi[F::NPAIR];
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                lo[p] = POW2 ? pw[p].x : __builtin_amdgcn_sqrtf(pw[p].x);
                hi[p] = POW2 ? pw[p].y : __builtin_amdgcn_sqrtf(pw[p].y);
            }
            const unsigned alo = lds_offset_of(prow + t), ahi = lds_offset_of(prow + (NC - t - 7 * F::LPF));
            asm volatile("ds_write2st64_b32 %8, %0, %1 offset0:0 offset1:1\n\t"
                         "ds_write2st64_b32 %8, %2, %3 offset0:2 offset1:3\n\t"
                         "ds_write2st64_b32 %8, %4, %5 offset0:4 offset1:5\n\t"
                         "ds_write2st64_b32 %8, %6, %7 offset0:6 offset1:7"
                         :: "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]), "v"(lo[7]), "v"(alo)
                         : "memory");
            asm volatile("ds_write2st64_b32 %8, %0, %1 offset0:7 offset1:6\n\t"
                         "ds_write2st64_b32 %8, %2, %3 offset0:5 offset1:4\n\t"
                         "ds_write2st64_b32 %8, %4, %5 offset0:3 offset1:2\n\t"
                         "ds_write2st64_b32 %8, %6, %7 offset0:1 offset1:0"
                         :: "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]), "v"(ahi)
                         : "memory");
        }
        if (t == 0) prow[NC / 2] = POW2 ? pmid : __builtin_amdgcn_sqrtf(pmid);
        if (t < C::PROW - NBINS) prow[NBINS + t] = 0.0f;                    // slack taps carry zero weights: keep them finite
        wave_lds_fence();
        // ---- the next frame's samples go out now (v is dead), they land during the contraction
        const int cur = i;
        i = (int)__builtin_amdgcn_readfirstlane(ask);
        request(i);
        // ---- s4: filterbank contraction, dB, row store
        if constexpr (FAST1 > 0) {
            const int ci = cur < nloc ? cur : nloc - 1;
            const f4* wp = reinterpret_cast<const f4*>(wlds) + lane;
            const f4* p0 = reinterpret_cast<const f4*>(prow + lo_s[0]);
            const f4* p1 = reinterpret_cast<const f4*>(prow + lo_s[1]);
            cf a0 = mkc(0.f, 0.f), a1 = mkc(0.f, 0.f), b0 = mkc(0.f, 0.f), b1 = mkc(0.f, 0.f);
            constexpr int B1 = (FAST1 + 1) / 2, B2 = FAST1 - B1;              // slot 1 in two batches
            f4 w0[ST_FAST_STEPS0], q0[ST_FAST_STEPS0], wa[B1], qa[B1];
#pragma unroll
            for (int u = 0; u < ST_FAST_STEPS0; ++u) {
                w0[u] = wp[u * 64];
                q0[u] = p0[u];
            }
#pragma unroll
            for (int u = 0; u < B1; ++u) {
                wa[u] = wp[(ST_FAST_STEPS0 + u) * 64];
                qa[u] = p1[u];
            }
#pragma unroll
            for (int u = 0; u < ST_FAST_STEPS0; ++u) fma4(w0[u], q0[u], a0, a1);
            f4 wb[B2], qb[B2];
#pragma unroll
            for (int u = 0; u < B2; ++u) {
                wb[u] = wp[(ST_FAST_STEPS0 + B1 + u) * 64];
                qb[u] = p1[B1 + u];
            }
#pragma unroll
            for (int u = 0; u < B1; ++u) fma4(wa[u], qa[u], b0, b1);
#pragma unroll
            for (int u = 0; u < B2; ++u) fma4(wb[u], qb[u], b0, b1);
            float v0 = (a0.x + a0.y) + (a1.x + a1.y), v1 = (b0.x + b0.y) + (b1.x + b1.y);
            if (m.db) {
                v0 = fast_db ? amp_to_db_fast(v0, m.amin, ten_log10_ref) : amp_to_db(v0, m.amin, m.log10_ref);
                v1 = fast_db ? amp_to_db_fast(v1, m.amin, ten_log10_ref) : amp_to_db(v1, m.amin, m.log10_ref);
            }
            float* orow = m.out + (begin + place(ci)) * (long long)m.n_mels + lane;
            orow[0] = v0;
            orow[64] = v1;
        } else {
            const int ci = cur < nloc ? cur : nloc - 1;
            const f4* wp = reinterpret_cast<const f4*>(wlds) + lane;
            float* orow = m.out + (begin + place(ci)) * (long long)m.n_mels + (m.rev ? m.n_mels - 1 - lane : lane);
            const int ostep = m.rev ? -64 : 64;                           // cell 64 s + lane -> its band (StreamArgs::rev)
#pragma unroll
            for (int s = 0; s < ST_MAX_SLOTS; ++s) {
                if (s < m.nslot) {
                    const f4* pp = reinterpret_cast<const f4*>(prow + lo_s[s]);
                    const int n = m.steps[s];
                    cf acc0 = mkc(0.f, 0.f), acc1 = mkc(0.f, 0.f);
                    int j = 0;
#pragma unroll 1
                    for (; j + 8 <= n; j += 8, wp += 512) {
                        f4 wv[8], pv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            wv[u] = wp[u * 64];
                            pv[u] = pp[j + u];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) fma4(wv[u], pv[u], acc0, acc1);
                    }
                    if (j + 4 <= n) {
                        f4 wv[4], pv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            wv[u] = wp[u * 64];
                            pv[u] = pp[j + u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) fma4(wv[u], pv[u], acc0, acc1);
                        wp += 256;
                        j += 4;
                    }
#pragma unroll 1
                    for (; j < n; ++j, wp += 64) fma4(wp[0], pp[j], acc0, acc1);
                    float val = (acc0.x + acc0.y) + (acc1.x + acc1.y);
                    if (m.db) val = fast_db ? amp_to_db_fast(val, m.amin, ten_log10_ref) : amp_to_db(val, m.amin, m.log10_ref);
                    if (s * 64 + lane < m.n_mels) orow[s * ostep] = val;
                }
            }
        }
        wave_lds_fence();                                                   // the row is consumed: the area takes the next frame
    }
    if (m.probe && w == 0) {
        const unsigned long long dc = __builtin_readcyclecounter() - probe_c, dw = wall_clock64() - probe_w;
        if (lane == 0) {
            m.probe[2 * blockIdx.x] = dc;
            m.probe[2 * blockIdx.x + 1] = dw;
        }
    }
}

}  // namespace tac
