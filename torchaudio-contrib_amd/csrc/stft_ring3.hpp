// stft_ring3.hpp — the fft_length-2048 STFT / spectrogram rows with the samples in an LDS HOP RING (round 5).
//
// hop = fft_length / 4: every sample belongs to four frames.  stft_stream3_kernel lets each frame load its 2048 samples from
// global memory — the 4x overlap is served by L1 / L2, no HBM byte is read twice, but three quarters of the kernel's vector-memory
// instructions and of its L2 -> CU bytes (as many bytes as the complex rows store) are re-reads, on a kernel that runs at the
// board's power limit and whose loads queue behind its own row stores.  Loading only a frame's NEW hop measures -6 ... -12 %
// (timing-only ablation, profiles/r05/ab/batch9_ab_newhop.txt); keeping the shared hops in a wave's registers (consecutive frames
// per wave) gives that back through its output pattern (batch10 / 11).  This form keeps one frame per wave with a CU's waves on
// ADJACENT frames (so that a CU — and, in the complex rows' global block order, the whole grid — writes one contiguous region)
// and shares the samples through the LDS instead:
//
//   * a workgroup is TW transform waves + ONE loader wave.  The loader walks the workgroup's frames (its contiguous chunk — real
//     rows — or its blocks of the global order — complex rows, see KB in the kernel) and brings every hop
//     (512 samples = 2 KB) an interior frame needs into a ring of R slots, once, with gfx950's LDS-DMA loads
//     (global_load_lds_dwordx4: global -> LDS without registers, lane l's 16 bytes land at M0 + 16 l; tools/ubench/lds_dma.hip) —
//     PF hops in flight, published in order (in-order vmcnt) through one LDS word `loaded`.  The loader counts its loads itself; the
//     compiler must not add memory waits of its own in its loops (it does when the register allocator reuses a pending load's
//     address register: +106 % on the kernel) — tests/test_host_api.py checks the built assembly for exactly the expected waits;
//   * hop h of audio row r has the id r * (T + 4) + h (block order: 15 dense ids per block of twelve frames) and lives in slot
//     id mod R; the frame starting at hop h0 takes ids B .. B + 3 (B = r * (T + 4) + h0), waits for loaded > B + 3, reads its sixteen complex pairs per lane with ds_read_b64
//     (conflict-free: consecutive lanes, consecutive 8 bytes) and marks itself consumed; the transform waves issue no global load
//     at all (frames touching the padding still gather theirs from memory, as before);
//   * the loader overwrites a slot only when every frame that can need its old content has consumed: ids below B(c), c = the first
//     frame of the chunk not yet consumed (it scans the per-frame marks in order); it never runs more than R hops ahead of B(c),
//     which is also what bounds the marks' ring.
// The first unconsumed frame can always get its hops (R >= 4 + waves in flight), so the protocol cannot deadlock; waits are
// bounded anyway, and a bound that expires traps (ring3_wait).  Shipped for every row mode (complex, |X|^2, |X|, dB): same process against stft_stream3_kernel complex rows
// -6 ... -9 % and a further -3.8 % from the nontemporal hop loads, real rows -8.3 % (12 + 1 waves with the ring against 16 without),
// bit-identical (profiles/r05/ab/batch14, batch17, batch19).
// Replaces torch.stft (reference functional.py:99-107) [+ complex_norm (functional.py:126-128)] [+ amplitude_to_db (291-296)].
#pragma once
#include "stft_stream3.hpp"

namespace tac {

// HPF: hops per frame (fft_length / hop): 4 (the benchmark's 2048 / 512) or 8 (2 does not fit: thirteen 4 KB hops beside twelve areas)
template <int NC, int E, int MODE, int TW, int HPF = 4>
struct Ring3Cfg {
    using F = WaveFft<NC, E>;
    static_assert(HPF == 4 || HPF == 8, "hop = fft_length / 4 or / 8");
    static constexpr int HOP = 2 * NC / HPF;                                 // samples per hop
    static constexpr int HOPB = HOP * 4;                                     // bytes per ring slot
    static constexpr int XA = s3_xa_bytes<F>();
    static constexpr int TABLES = ST_TW_BYTES + 64 + 64 * (F::NPAIR + E) * (int)sizeof(cf);
    static constexpr int MARKS = 64;                                         // ring of per-frame "consumed" marks
    static constexpr int CTRL = MARKS * 4 + 64;                              // + loaded, front, frame counter
    static constexpr int LDS_MAX = 160 * 1024;
    static constexpr int R_LDS = (LDS_MAX - TW * XA - TABLES - CTRL) / HOPB;     // hops the LDS has room for
    static constexpr int R = R_LDS < MARKS - TW - 4 ? R_LDS : MARKS - TW - 5;    // hops in the ring (bounded by the marks' ring)
    static constexpr int BYTES = R * HOPB + TW * XA + TABLES + CTRL;
    // hops the loader keeps in flight: as far ahead as the ring allows beyond the frames being transformed (vmcnt counts to 63)
    static constexpr int PF = R - TW - (HPF - 1) > 24 ? 24 : R - TW - (HPF - 1);
    static constexpr int LPH = HOPB / 1024;                                  // LDS-DMA instructions (1 KB each) per hop
    static_assert(R >= TW + HPF + 2 && R < MARKS - TW - 4 && BYTES <= LDS_MAX && PF >= 1 && LPH * (PF - 1) <= 63, "ring");
};

// word >= want (wrap-safe), polled by the whole wave.  Bounded — ~0.1 s at full clock, far beyond anything a profiler or a throttled
// clock stretches a hop load to —, and a bound that expires is a protocol failure: the wave TRAPS (s_trap 2, what llvm.trap lowers
// to: the queue goes into its error state, the host sees hipErrorLaunchFailure at its next synchronisation) instead of going on
// with a ring it may not read — garbage rows with no error code are not an outcome.  (As inline assembly: __builtin_trap() is
// noreturn, and the changed control flow cost this register-tight kernel 8 % on the |X|^2 rows — profiles/r06/ab/batch7.)
__device__ __forceinline__ void ring3_wait(unsigned addr, unsigned want) {
    bool ok = false;
    for (int guard = 0; guard < (1 << 20); ++guard) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        if ((int)(v - want) >= 0) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) asm volatile("s_trap 2");
}

// MODE as in stft_stream3_kernel.  Launch conditions (host): hop == fft_length / 4, center_pad a multiple of the hop, 16-byte
// aligned hops (FrameGeom::vec4_ok), rows * (T + 4) < 2^31, rows of at least two frames.
template <int NC, int E, int MODE, int TW, int HPF = 4>
__global__ void __launch_bounds__((TW + 1) * 64, (TW + 1 + 3) / 4)
stft_ring3_kernel(FrameGeom g, Tables tb, StftEpilogue ep, Stream3Launch lp) {
    using F = WaveFft<NC, E>;
    using D = Ring3Cfg<NC, E, MODE, TW, HPF>;
    static_assert(F::G == 1 && E == 16 && radix_at(NC, 0) == 16, "fft_length 2048");
    constexpr int WAVES = TW + 1;
    constexpr int XA_BYTES = D::XA;
    constexpr int R = D::R;
    constexpr int LENF = (MODE == 0 ? 2 : 1) * (NC + 1);
    constexpr int NST = ((LENF >> 2) + 63) / 64;          // 16-byte wave-stores per output row
    static_assert(XA_BYTES >= (LENF + 3) * 4, "the staged row (any 16-byte phase) fits the exchange area");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const ring = smem_raw;
    unsigned char* const areas = smem_raw + (size_t)R * D::HOPB;
    float* const twlds = reinterpret_cast<float*>(areas + (size_t)TW * XA_BYTES);
    unsigned* const next_frame = reinterpret_cast<unsigned*>(twlds + ST_TW_BYTES / 4);
    cf* const ptwl = reinterpret_cast<cf*>(next_frame + 16);
    cf* const winl = ptwl + 64 * F::NPAIR;
    unsigned* const marks = reinterpret_cast<unsigned*>(winl + 64 * E);       // marks[i mod 64] = i + 1: frame i has its samples
    unsigned* const loaded = marks + D::MARKS;                                // ids below this are in the ring
    unsigned* const front = loaded + 1;                                       // frames below this have consumed
    const float half = 0.5f * g.scale;
    S3Setup<F, WAVES * 64> setup;
    setup.issue(g, tb, tid);

    // KB > 0: GLOBAL BLOCK ORDER (round 5, late) — an audio row is cut into blocks of KB consecutive frames, the blocks of all rows
    // are numbered in output order and block q goes to workgroup slot q mod G at turn q / G, so that at any time the whole grid
    // writes ONE tight window of G * KB adjacent rows of the output (0: every workgroup walks its own contiguous chunk — 256
    // far-apart streams).  The store pattern alone gains 3 - 7 % (8 200-byte rows) from it (tools/ubench/row_store_rate.hip);
    // the complex-row kernel, bit-identical, -2.4 ... -3.9 % on the boxes where it runs slowest (0.171 - 0.175 ms) and
    // +0.3 ... +1.5 % on the fastest (0.157 - 0.164): shipped for the complex rows.  The real rows stay in chunk order: a block
    // reads 15 hops for 12 frames and their loader, whose one LDS round trip per hop is just under the frame interval there, cannot
    // take 25 % more (+13.5 %; profiles/r05/ab/batch50 ... 54).
    // (hop = fft_length / 8 would read 19 hops per 12 frames: chunk order too)
    constexpr int KB = MODE == 0 && HPF == 4 ? 12 : 0, PB = KB + HPF - 1;     // frames per block, hop ids per block (dense: KB + HPF - 1 hops are read)
    const long long total = g.rows * g.n_frames;
    const long long chunk = lp.chunk;
    const long long begin = (long long)blockIdx.x * chunk;
    const long long endl = begin + chunk < total ? begin + chunk : total;
    const unsigned T = (unsigned)g.n_frames;
    const unsigned G = gridDim.x;
    // workgroups of one XCD (blockIdx % 8) side by side: every XCD's L2 sees one contiguous part of the window
    const unsigned slot = (G & 7u) == 0 ? (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned nb = KB ? (T + (unsigned)(KB ? KB : 1) - 1) / (unsigned)(KB ? KB : 1) : 0;      // blocks per audio row
    const unsigned nblocks = (unsigned)g.rows * nb;
    const int nloc = KB ? (slot < nblocks ? (int)((nblocks - slot + G - 1) / G) * KB : 0) : (endl > begin ? (int)(endl - begin) : 0);
    const unsigned HR = T + (unsigned)HPF;                // ids per audio row (chunk order)
    const int padh = g.center_pad / D::HOP;               // hops of padding in front of frame 0
    // frame i of the workgroup: (row, frame in the row — >= T: a hole behind a row's last block —, interior?, id of its first hop
    // — also a lower bound for edge frames)
    auto locate = [&](int i, unsigned& r, unsigned& f, bool& ok, int& b) {
        if constexpr (KB > 0) {
            const unsigned j = (unsigned)i / (unsigned)KB, k = (unsigned)i - j * (unsigned)KB;
            const unsigned q = j * G + slot;
            r = q / nb;
            f = (q - r * nb) * (unsigned)KB + k;
            b = (int)(j * (unsigned)PB + k);
        } else {
            const unsigned gf = (unsigned)(begin + i);
            r = gf / T;
            f = gf - r * T;
            b = (int)(r * HR) + (int)f - padh;
        }
        const long long start = (long long)f * D::HOP - g.center_pad;
        ok = f < T && start >= 0 && start + F::N <= g.length;
    };
    typedef float f4 __attribute__((ext_vector_type(4)));

    if (tid < D::CTRL / 4) marks[tid] = 0u;
    if (tid == 0) *next_frame = TW;
    setup.store(twlds, ptwl, winl, half, tid);
    __syncthreads();
    if (nloc <= 0) return;

    if (w == TW) {
        // =================================================== the loader
        __builtin_amdgcn_s_setprio(3);
        const unsigned marks_addr = lds_offset_of(marks), loaded_addr = lds_offset_of(loaded), front_addr = lds_offset_of(front);
        typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
        uint4_t mq;
        int c = 0, bc;                                    // first frame not known to have consumed, the id bound it implies
        {
            unsigned r, f;
            bool ok;
            locate(0, r, f, ok, bc);
        }
        // frames [c, ...) that carry their mark become consumed; returns whether c moved.  One LDS round trip reads the aligned group
        // of four marks c belongs to; the scan goes on while whole groups are consumed (at most four groups per call).
        auto advance = [&]() {
            bool moved = false;
            for (int step = 0; step < 4 && c < nloc; ++step) {
                const int c0 = c & ~3;
                unsigned m0, m1, m2, m3;
                const unsigned a = marks_addr + 4u * (unsigned)(c0 & (D::MARKS - 1));
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(mq) : "v"(a) : "memory");
                m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)mq.x);
                m1 = (unsigned)__builtin_amdgcn_readfirstlane((int)mq.y);
                m2 = (unsigned)__builtin_amdgcn_readfirstlane((int)mq.z);
                m3 = (unsigned)__builtin_amdgcn_readfirstlane((int)mq.w);
                const int before = c;
                if (c == c0 && c < nloc && m0 == (unsigned)c + 1u) ++c;
                if (c == c0 + 1 && c < nloc && m1 == (unsigned)c + 1u) ++c;
                if (c == c0 + 2 && c < nloc && m2 == (unsigned)c + 1u) ++c;
                if (c == c0 + 3 && c < nloc && m3 == (unsigned)c + 1u) ++c;
                moved = moved || c != before;
                if (c != c0 + 4) break;                   // the group still holds an unconsumed frame
            }
            if (moved) {
                if (c < nloc) {
                    unsigned r, f;
                    bool ok;
                    locate(c, r, f, ok, bc);
                } else {
                    bc = 0x7fffffff - R;
                }
                const unsigned cv = (unsigned)c;
                asm volatile("ds_write_b32 %0, %1" :: "v"(front_addr), "v"(cv) : "memory");
            }
            return moved;
        };
        int fifo[D::PF];                                  // ids of the hops in flight, oldest first
#pragma unroll
        for (int k = 0; k < D::PF; ++k) fifo[k] = -1;
        int next_id = -0x7fffffff;
        for (int i = 0; i < nloc; ++i) {
            unsigned r, f;
            bool ok;
            int b;
            locate(i, r, f, ok, b);
            if (!ok) continue;
            for (int id = next_id > b ? next_id : b; id <= b + HPF - 1; ++id) {
                if (id - R >= bc) advance();
                if (id - R >= bc) {                       // the slot still holds a hop somebody needs
                    // ... and that somebody may be waiting for one of the hops still in flight: everything issued is published
                    // before the loader sleeps
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    if (fifo[D::PF - 1] >= 0) {
                        const unsigned lv = (unsigned)fifo[D::PF - 1] + 1u;
                        asm volatile("ds_write_b32 %0, %1" :: "v"(loaded_addr), "v"(lv) : "memory");
                    }
                    for (int guard = 0; id - R >= bc && guard < (1 << 20); ++guard) {
                        if (!advance()) __builtin_amdgcn_s_sleep(2);
                    }
                    if (id - R >= bc) asm volatile("s_trap 2");    // (a slot somebody still needs is never overwritten: see ring3_wait)
                }
                const unsigned slot = (unsigned)id % (unsigned)R;
                const int h = KB > 0 ? (int)f - padh + (id - b) : id - (int)(r * HR);  // hop of the row
                const float* src = g.wave + (long long)r * g.row_stride + (long long)h * D::HOP + 4 * t;
                unsigned char* dst = ring + (size_t)slot * D::HOPB;
                // cache policy 2 = nontemporal: a hop is read once per launch — the next CU's chunk shares nothing with this one —
                // so it need not stay in the L2 / Infinity Cache the row stores are streaming through: -3.8 % same process against
                // the default policy, sc0 +2.0 % (profiles/r05/ab/batch17)
#pragma unroll
                for (int k = 0; k < D::LPH; ++k)
                    __builtin_amdgcn_global_load_lds(src + 256 * k, (__attribute__((address_space(3))) void*)(dst + 1024 * k), 16, 0, 2);
                // the hop issued PF - 1 hops ago has landed (loads complete in order): publish it
#pragma unroll
                for (int k = 0; k + 1 < D::PF; ++k) fifo[k] = fifo[k + 1];
                fifo[D::PF - 1] = id;
                __builtin_amdgcn_s_waitcnt(0x0F70 | ((D::LPH * (D::PF - 1)) & 15) | (((D::LPH * (D::PF - 1)) >> 4) << 14));     // vmcnt(LPH (PF - 1))
                if (fifo[0] >= 0) {
                    const unsigned lv = (unsigned)fifo[0] + 1u;
                    asm volatile("ds_write_b32 %0, %1" :: "v"(loaded_addr), "v"(lv) : "memory");
                }
            }
            next_id = b + HPF;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): everything issued is in the ring
        if (fifo[D::PF - 1] >= 0) {
            const unsigned lv = (unsigned)fifo[D::PF - 1] + 1u;
            asm volatile("ds_write_b32 %0, %1" :: "v"(loaded_addr), "v"(lv) : "memory");
        }
        // keep the front moving for edge frames that wait for it (they never run more than a marks' ring ahead)
        for (int guard = 0; c < nloc && guard < (1 << 20); ++guard) {
            if (!advance()) __builtin_amdgcn_s_sleep(8);
        }
        return;
    }

    // ======================================================= transform waves
    cf* const xa = reinterpret_cast<cf*>(areas + (size_t)w * XA_BYTES);
    cf tw2[3];
    {
        cf all[F::NTW];
        F::load_twiddles(all, tb.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw2[q] = all[twiddles_before(NC, E, 2) + q];
    }
    const unsigned marks_addr = lds_offset_of(marks), loaded_addr = lds_offset_of(loaded), front_addr = lds_offset_of(front);
    const unsigned ring_lane = lds_offset_of(ring) + 8u * (unsigned)t;       // this lane's pair of a hop's first 64
    S3Swz swz;
    swz.init(xa, t);
    cf v[E];
    int i = w;
    while (i < nloc) {
        unsigned ask = 0;
        if (t == 0) ask = __hip_atomic_fetch_add(next_frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned row, fru;
        bool ok;
        int b;
        locate(i, row, fru, ok, b);
        const long long fr = (long long)fru;
        if (KB > 0 && fru >= T) {                           // a hole behind the last block of an audio row: nothing to do
            if (i >= D::MARKS) ring3_wait(front_addr, (unsigned)(i - D::MARKS) + 1u);
            const unsigned hm_addr = marks_addr + 4u * (unsigned)(i & (D::MARKS - 1)), hm = (unsigned)i + 1u;
            asm volatile("ds_write_b32 %0, %1" :: "v"(hm_addr), "v"(hm) : "memory");
            i = (int)__builtin_amdgcn_readfirstlane(ask);
            continue;
        }
        const long long g0 = (KB > 0 ? (long long)row * T + fr : begin + i) * (long long)LENF;     // this frame's row in the frame-major output
        const unsigned mark_addr = marks_addr + 4u * (unsigned)(i & (D::MARKS - 1)), mark = (unsigned)i + 1u;
        // ---- s0: the samples (ring, or gathered from memory for frames touching the padding), window, pass 0, exchange
        if (ok) {
            unsigned sa[HPF];                               // byte address of this lane's first pair in each of the frame's hops
            {
                unsigned s = (unsigned)b % (unsigned)R;
#pragma unroll
                for (int j = 0; j < HPF; ++j) {
                    sa[j] = ring_lane + s * (unsigned)D::HOPB;
                    s = s + 1 == (unsigned)R ? 0u : s + 1;
                }
            }
            // element m = t + 64 q of the frame: hop (q HPF) / 16, pair t + 64 (q mod (16 / HPF)) of it
            auto rd = [&](auto qc) { constexpr int q = decltype(qc)::value; v[q] = lds_read_b64_single<512 * (q % (16 / HPF))>(sa[(q * HPF) >> 4]); };
            // the loader's progress is read WITH the samples (one round trip instead of two; LDS operations complete in order, so a
            // `loaded` that covers the frame means the reads behind it saw the hops)
            unsigned have;
            asm volatile("ds_read_b32 %0, %1" : "=v"(have) : "v"(loaded_addr) : "memory");
            rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
            rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
            rd(std::integral_constant<int, 8>{}); rd(std::integral_constant<int, 9>{}); rd(std::integral_constant<int, 10>{}); rd(std::integral_constant<int, 11>{});
            rd(std::integral_constant<int, 12>{}); rd(std::integral_constant<int, 13>{}); rd(std::integral_constant<int, 14>{}); rd(std::integral_constant<int, 15>{});
            lds_wait_all(v);
            asm volatile("" : "+v"(have));                  // (its read completed with the others: no use may move above the wait)
            if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)have) - ((unsigned)b + (unsigned)HPF)) < 0) {
                // (rare: the loader was not that far yet when the reads above were issued — wait, read again)
                ring3_wait(loaded_addr, (unsigned)b + (unsigned)HPF);
                rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
                rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
                rd(std::integral_constant<int, 8>{}); rd(std::integral_constant<int, 9>{}); rd(std::integral_constant<int, 10>{}); rd(std::integral_constant<int, 11>{});
                rd(std::integral_constant<int, 12>{}); rd(std::integral_constant<int, 13>{}); rd(std::integral_constant<int, 14>{}); rd(std::integral_constant<int, 15>{});
                lds_wait_all(v);
            }
            asm volatile("ds_write_b32 %0, %1" :: "v"(mark_addr), "v"(mark) : "memory");
        } else {
            // (an edge frame takes nothing from the ring; it only must not lap the ring of marks)
            if (i >= D::MARKS) ring3_wait(front_addr, (unsigned)(i - D::MARKS) + 1u);
            asm volatile("ds_write_b32 %0, %1" :: "v"(mark_addr), "v"(mark) : "memory");
            int tz;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(t));
            load_frame<F, false, true, true>(v, g, nullptr, xa, (int)row, fr, tz, FetchF32{g.wave});
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
        F::template pass_twiddle<2, true>(v, tw2);
        F::template pass_butterflies<2>(v);
        cf zm[F::NPAIR], zmid;
        s3_r2c_partners<F>(v, xa, zm, zmid, t);
        // ---- s3: R2C split; the row overwrites the exchange area once every lane holds its partners
        cf ptw[F::NPAIR];
        {
            const f4* pl = reinterpret_cast<const f4*>(ptwl) + t;
#pragma unroll
            for (int u = 0; u < F::NPAIR / 2; ++u) {
                const f4 x = pl[u * 64];
                ptw[2 * u] = mkc(x.x, x.y);
                ptw[2 * u + 1] = mkc(x.z, x.w);
            }
        }
        const int a = (int)(g0 & 3);
        float* const stage = reinterpret_cast<float*>(xa) + a;      // LDS and global share their 16-byte phase
        if constexpr (MODE == 0) {
            cf xlo[F::NPAIR], xhi[F::NPAIR], xm, unused;
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) F::r2c_split_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], xlo[p], xhi[p]);
            F::r2c_split_x2(zmid, zmid, mkc(0.0f, -1.0f), xm, unused);
            wave_lds_fence();
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int kk = t + p * F::LPF;
                reinterpret_cast<cf*>(stage)[kk] = xlo[p];
                reinterpret_cast<cf*>(stage)[NC - kk] = xhi[p];
            }
            if (t == 0) reinterpret_cast<cf*>(stage)[NC / 2] = xm;
        } else {
            cf pw[F::NPAIR];
#pragma unroll
            for (int p = 0; p < F::NPAIR; p += 2)
                r2c_power_pair_x2(v[F::reg_of_spectrum(p)], zm[p], ptw[p], v[F::reg_of_spectrum(p + 1)], zm[p + 1], ptw[p + 1], pw[p], pw[p + 1]);
            const float pmid = 4.0f * cnorm2(zmid);
            wave_lds_fence();
#pragma unroll
            for (int p = 0; p < F::NPAIR; ++p) {
                const int kk = t + p * F::LPF;
                stage[kk] = spectral_row_value<MODE>(pw[p].x, ep);
                stage[NC - kk] = spectral_row_value<MODE>(pw[p].y, ep);
            }
            if (t == 0) stage[NC / 2] = spectral_row_value<MODE>(pmid, ep);
        }
        wave_lds_fence();
        i = (int)__builtin_amdgcn_readfirstlane(ask);
        // ---- the row leaves as 1 + NST + 1 unconditional stores: out-of-range lanes repeat a neighbour's element
        float* const gdst = ep.out + g0;
        const int npre = (4 - a) & 3;
        const int nchunks = (LENF - npre) >> 2;
        {
            const int hmax = (npre > 1 ? npre : 1) - 1;
            const int hi = t < hmax ? t : hmax;
            gdst[hi] = stage[hi];
        }
        {
            const f4* const s4 = reinterpret_cast<const f4*>(stage + npre);
            f4* const g4 = reinterpret_cast<f4*>(gdst + npre);
            const int last = nchunks - 1;
            f4 bb[NST];
            int c[NST];
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int j = t + 64 * u;
                c[u] = j < last ? j : last;
                bb[u] = s4[c[u]];
            }
            __builtin_amdgcn_sched_barrier(0);            // all LDS reads in flight before the first store issues
            if (lp.plain_stores) {
#pragma unroll
                for (int u = 0; u < NST; ++u) g4[c[u]] = bb[u];
            } else {
#pragma unroll
                for (int u = 0; u < NST; ++u) __builtin_nontemporal_store(bb[u], &g4[c[u]]);
            }
        }
        {
            const int r = LENF - npre - 4 * nchunks;
            const int rmax = (r > 1 ? r : 1) - 1;
            const int ti = LENF - 1 - (t < rmax ? t : rmax);
            gdst[ti] = stage[ti];
        }
        wave_lds_fence();                                 // the next frame's first-pass writes follow these reads
    }
}

}  // namespace tac
