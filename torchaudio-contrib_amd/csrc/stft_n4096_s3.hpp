// stft_n4096_s3.hpp — the real-valued rows (|X|, |X|^2, their dB forms) of fft_length 4096 with THREE waves per SIMD.
//
// stft_n4096_kernel (stft_n4096.hip) keeps both 1024-point transforms of a frame alive at once: two 8.7 KB exchange areas and ~250
// registers per wave, i.e. eight waves per CU, and its counters say what that costs (profiles/r05/pmc_spec4096.json: VALU 56 % busy,
// waves parked 28 % of their cycles, ~880 LDS cycles per frame of which a third are the padded layout's conflicts and hipcc's merged
// read2 pairs).  This form runs the one-frame-per-wave front end of the fft_length-2048 kernels (melspec_stream3.hpp: windowed first
// butterfly, XOR-swizzled first exchange read back as single b64s, pass 1 -> 2 in registers, dense partner exchange, tables in LDS)
// TWICE through ONE area — A = FFT_1024(z[2m]), then B = FFT_1024(z[2m+1]) — keeping of each only what the combine needs (its
// lower half and the partners of that half: 34 registers), combines each pair (k, 1024 - k) once into its four bins as
// stft_n4096_kernel does, and stages the 2049-float row in place over the area: twelve <= 168-register waves per CU.  Round 3's
// twelve-wave attempt (tools/ablation/stft_n4096_one_area_r03.patch) had the hoisted twiddles of the old front end in registers
// and no room to request the next frame before the current one was done; here the transforms need ~100 registers beside the 34 of
// the first one's result, and the sixteen 16-byte requests of the NEXT frame go out once the row is staged, ahead of the row stores
// (three waves per SIMD cover them: with every load served from cache the kernel is no faster, tools/ablation/README.md "Round 6").
// With MEL the staged |X|^p row is not stored but contracted with a band-sparse filterbank in place (lane l owns bands l, 64 + l, ...;
// every band slot runs ITS OWN number of four-tap steps from a table in LDS) and only the mel (dB) row leaves: Melspectrogram
// (-> AmplitudeToDb) at fft_length 4096 in ONE launch (reference layers.py:307-381).  Replaces torch.stft + complex_norm [+ amplitude_to_db] (reference functional.py:99-107, 126-128, 291-296)
// for layers.py:267-304 at fft_length 4096.
#pragma once
#include "stft_stream3.hpp"

namespace tac {

constexpr int N4S_WAVES = 12;
typedef float n4s_f4 __attribute__((ext_vector_type(4)));

// exchange areas + pass-1 twiddle sets + frame counter + W_2048^k pairs + the two window tables (even / odd complex elements)
__host__ __device__ constexpr size_t n4096_s3_lds_bytes(int waves) {
    return (size_t)waves * s3_xa_bytes<WaveFft<1024, 16>>() + ST_TW_BYTES + 64 + 64 * 8 * sizeof(cf) + 2 * 64 * 16 * sizeof(cf);
}
// ... + the row-store form's twiddle tables (pass 2: 64 x 12, W_4096^k: 64 x 8 complex values)
constexpr size_t N4S_ROW_TABLE_BYTES = 64 * 20 * sizeof(cf);

// the filterbank of the fused form (tac_melbank_pack for fft_length 4096; built by pack_n4096_mel, stft_n4096.hip).  A band's run of
// four-tap steps is cut into PIECES of about equal length; the pieces of all bands, longest first, fill the cells (slot, lane) of up to
// N4M_SLOTS slots row by row, and slot s runs the step pairs of ITS longest piece — so that 128 mel bands over 2049 bins cost ~20 steps
// per frame instead of the 40 of "lane l owns bands l and 64 + l", and banks with few, wide bands (40 or 80 bands: up to 300 bins) fit at
// all.  A lane keeps its slots' partial sums in registers; once the row is dead they go to the area as cells [slot][lane], and band
// 64 r + l is the sum of the `np` cells (a multiple of four) its mix list names (padded with the index of a cell that holds zero).
// When no band needs cutting for the table to fit (np == 0: "direct") cell c = 64 r + l IS band c — or band n_mels - 1 - c (`rev`), so
// that the widest bands of a bank whose band count is not a multiple of 64 share a slot: the lane that contracted it applies the dB and
// stores it (one coalesced 256-byte store per slot), nothing is gathered and there is no mix table.
//   desc: int32 first bins [N4M_SLOTS][64], step pairs [N4M_SLOTS] (0: slot not used), 10 ints of padding, mix [rounds][np][64]
//   wpack: float [step][lane][4 taps], a slot's steps behind the previous slot's
constexpr int N4M_SLOTS = 6;
constexpr int N4M_MAX_PIECES = 16;
constexpr int N4M_MAX_MELS = 256;
constexpr int N4M_DESC_HEAD = 64 * N4M_SLOTS + N4M_SLOTS + 10;            // ints in front of the mix table (a multiple of four)
constexpr int N4M_ZERO_CELL = 64 * N4M_SLOTS;
constexpr int N4M_MARK = 1000 + 4096;                                      // info_host[2] of such a pack
struct N4Mel {
    const float* wpack;
    const int* desc;
    int np, rounds, wtot, n_mels, db, rev;
    float amin, log10_ref;
    float* out;                 // [rows][T][n_mels]
};
__host__ __device__ constexpr int n4096_mix_rows(int rounds, int np) { return rounds * np; }
__host__ __device__ constexpr size_t n4096_mel_lds_bytes(int rounds, int np, int wtot) {
    return (size_t)(N4M_DESC_HEAD + 64 * n4096_mix_rows(rounds, np)) * sizeof(int) + (size_t)wtot * sizeof(float);
}

// r2c_power_pair_x2 (fft_core.hpp) for the pairs (zk1, zm1, w) and (zk2, zm2, -i conj(w)): the second twiddle is never formed, its
// components are picked out of w by the operand modifiers of the two instructions that use it
__device__ __forceinline__ void r2c_power_pair_x2_mirror(cf zk1, cf zm1, cf zk2, cf zm2, cf w, cf& p1, cf& p2) {
    cf ev1, ev2, d1, d2, t1, t2, tw1, tw2, re1, re2, im1, im2, q1, q2;
    asm("v_pk_add_f32 %0, %16, %17 neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %18, %19 neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %16, %17 neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %3, %18, %19 neg_lo:[0,1]\n\t"
        "v_pk_mul_f32 %4, %2, %20 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %5, %3, %20 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %6, %2, %20, %4 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %7, %3, %20, %5 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_add_f32 %8, %0, %6 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %9, %1, %7 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %10, %0, %6 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %11, %1, %7 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %12, %8, %8\n\t"
        "v_pk_mul_f32 %13, %9, %9\n\t"
        "v_pk_fma_f32 %14, %10, %10, %12\n\t"
        "v_pk_fma_f32 %15, %11, %11, %13"
        : "=&v"(ev1), "=&v"(ev2), "=&v"(d1), "=&v"(d2), "=&v"(t1), "=&v"(t2), "=&v"(tw1), "=&v"(tw2), "=&v"(re1), "=&v"(re2),
          "=&v"(im1), "=&v"(im2), "=&v"(q1), "=&v"(q2), "=&v"(p1), "=&v"(p2)
        : "v"(zk1), "v"(zm1), "v"(zk2), "v"(zm2), "v"(w));
}

// MODE: 1 |X|^2, 2 |X|, 3 |X|^2 in dB, 4 |X| in dB (spectral_row_value); MEL: the row is contracted with `mel` instead of stored
// (MODE 1 / 2 only); WAVES: 12, or fewer when the bank's table needs the LDS
// TBL: the pass-2 / W_4096 twiddle tables in LDS (always in the row-store form; in the fused form where the bank leaves the room)
template <int MODE, int WAVES, bool MEL, bool TBL = !MEL>
__global__ void __launch_bounds__(WAVES * 64, (WAVES + 3) / 4)
stft_n4096_s3_kernel(FrameGeom g, Tables tb2k, Tables tb4k, StftEpilogue ep, N4Mel mel) {
    constexpr int N4S_WAVES = WAVES;
    static_assert(MEL || TBL, "the row-store form has the room");
    using F = WaveFft<1024, 16>;
    using f4 = n4s_f4;
    static_assert(MODE >= 1 && MODE <= 4 && (!MEL || MODE <= 2), "real-valued rows");
    constexpr int E = 16, NCH = 1024, LENF = 2049;
    constexpr int XA_BYTES = s3_xa_bytes<F>();
    constexpr int NST = ((LENF >> 2) + 63) / 64;          // 16-byte wave-stores per output row: 9
    static_assert(XA_BYTES >= (LENF + 3) * 4, "the staged row (any 16-byte phase) fits the exchange area");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* const xa = reinterpret_cast<cf*>(smem_raw + (size_t)w * XA_BYTES);
    float* const twlds = reinterpret_cast<float*>(smem_raw + (size_t)N4S_WAVES * XA_BYTES);
    unsigned* const next_unit = reinterpret_cast<unsigned*>(twlds + ST_TW_BYTES / 4);
    cf* const ptwl = reinterpret_cast<cf*>(next_unit + 16);               // W_2048^(t + 64 p) as [p >> 1][lane][p & 1]
    cf* const winA = ptwl + 64 * 8;                                       // window pairs of z[2 (t + 64 q)] as [q >> 1][lane][q & 1]
    cf* const winB = winA + 64 * E;                                       // ... of z[2 (t + 64 q) + 1]
    int* const mlo = reinterpret_cast<int*>(winB + 64 * E);               // MEL: the pack's desc (first bins, step pairs, mix table)
    float* const mwl = reinterpret_cast<float*>(mlo + N4M_DESC_HEAD + 64 * n4096_mix_rows(mel.rounds, mel.np));

    // ---- tables (the R2C split returns 2X: the halving and the transform's scale are folded into the window)
    const float half = 0.5f * g.scale;
    for (int c = tid; c < S3_IMG_TW1_F4 + S3_IMG_PTW_F4; c += N4S_WAVES * 64) {
        const f4 x = reinterpret_cast<const f4*>(tb2k.s3img)[c];
        if (c < S3_IMG_TW1_F4) reinterpret_cast<f4*>(twlds)[c] = x;
        else reinterpret_cast<f4*>(ptwl)[c - S3_IMG_TW1_F4] = x;
    }
    for (int j = tid; j < NCH; j += N4S_WAVES * 64) {
        const int q = j >> 6, tt = j & 63;
        const int slot = ((q >> 1) * 64 + tt) * 2 + (q & 1);
        winA[slot] = cscale(window_pair(g, 2 * j), half);
        winB[slot] = cscale(window_pair(g, 2 * j + 1), half);
    }
    if constexpr (MEL) {
        for (int i = tid; i < N4M_DESC_HEAD + 64 * n4096_mix_rows(mel.rounds, mel.np); i += N4S_WAVES * 64) mlo[i] = mel.desc[i];
        for (int i = tid; i < (mel.wtot >> 2); i += N4S_WAVES * 64)
            reinterpret_cast<f4*>(mwl)[i] = reinterpret_cast<const f4*>(mel.wpack)[i];
    }
    // TBL (the row-store form always; round 6, late): the pass-2 twiddles with the last pass's constants multiplied in — W_1024^((t + 64 b) q), b < 4,
    // q = 1 .. 3, as [u < 6][lane] pairs — and the eight W_4096^(t + 64 p) come from LDS tables instead of being formed per frame from
    // four registers: 52 fewer vector instructions per frame (of ~1 050) and five fewer spilled registers for sixteen more 16-byte LDS reads;
    // same process -5.2 % on the cfg-4 slice, -6.3 % at full size (4.683 -> 4.390 ms = 50.4 % of 8 TB/s).  The kernel is bound by its
    // instruction stream at the power limit and its LDS is 40 % busy; the fused form (MEL) takes the tables where the bank leaves them the
    // room (TBL; behind the bank's weights), else it keeps the registers.
    cf* const tw2l = MEL ? reinterpret_cast<cf*>(mwl + mel.wtot) : reinterpret_cast<cf*>(mlo);
    if constexpr (TBL) {
        for (int i = tid; i < 64 * 12; i += N4S_WAVES * 64) {
            const int tt = i & 63, e = i >> 6, b = e / 3, q = e % 3 + 1;
            tw2l[((e >> 1) * 64 + tt) * 2 + (e & 1)] = tb2k.w_nc[((tt + 64 * b) * q) & 1023];
        }
    }
    cf* const w4l = tw2l + 64 * 12;                                        // W_4096^(t + 64 p), p < 8, as [p >> 1][lane][p & 1]
    if constexpr (TBL) {
        for (int i = tid; i < 64 * 8; i += N4S_WAVES * 64) {
            const int tt = i & 63, p8 = i >> 6;
            w4l[((p8 >> 1) * 64 + tt) * 2 + (p8 & 1)] = tb4k.w_n[tt + 64 * p8];
        }
    }
    cf tw2[3];
    {
        cf all[F::NTW];
        F::load_twiddles(all, tb2k.w_nc, t);
#pragma unroll
        for (int q = 0; q < 3; ++q) tw2[q] = all[twiddles_before(NCH, E, 2) + q];
    }
    const cf w4k = tb4k.w_n[t];                           // W_4096^t

    const int T = (int)g.n_frames;
    const int total = (int)g.rows * T;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total ? begin + chunk : total;

    // sixteen 16-byte requests per lane: samples 4m .. 4m+3, m = t + 64 q — element q of A (.xy) and of B (.zw).  Always issued, from
    // a start clamped into the row (host-checked: length >= 4096); `mode` says whether the registers are this frame's samples
    f4 raw[E];
    int mode = 0, row = 0, fr = 0;
    auto request = [&](int unit) {
        unit = unit < end ? unit : end - 1;
        row = unit / T;
        fr = unit - row * T;
        const long long start = (long long)fr * g.hop - g.center_pad;
        mode = (start >= 0 && start + 4096 <= g.length) ? 1 : 2;
        long long cs = start < 0 ? 0 : start;
        cs = cs + 4096 <= g.length ? cs : g.length - 4096;
        const f4* src = reinterpret_cast<const f4*>(g.wave + (long long)row * g.row_stride + (cs & ~3ll));
#pragma unroll
        for (int q = 0; q < E; ++q) raw[q] = src[t + 64 * q];
    };
    int unit = begin + w;
    if (begin < end) request(unit);
    __builtin_amdgcn_sched_barrier(0);
    if (tid == 0) *next_unit = (unsigned)(begin + N4S_WAVES);
    __syncthreads();
    if (unit >= end) return;

    S3Swz swz;
    swz.init(xa, t);
    // one 1024-point transform of the front end: v = raw samples of the sixteen first-pass elements; returns the lane's lower half
    // Z[t + 64 p] (lo), the partners Z[1024 - (t + 64 p)] (zm) and Z[512] (mid)
    auto transform = [&](cf (&v)[E], const cf* winl, cf (&lo)[8], cf (&zm)[8], cf& mid) {
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
        F::template pass_twiddle<1, true>(v, tw1);
        F::template pass_butterflies<1>(v);
        F::exchange_1_2_in_registers(v);
        if constexpr (TBL) {
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
        s3_r2c_partners<F>(v, xa, zm, mid, t);
#pragma unroll
        for (int p = 0; p < 8; ++p) lo[p] = v[F::reg_of_spectrum(p)];
        wave_lds_fence();
    };
    // frames touching the padding: the raw samples of one half-sequence gathered through the area (rolled), then the same path
    auto gather = [&](cf (&v)[E], int odd) {
        const float* rp = g.wave + (long long)row * g.row_stride;
        const int s0 = (int)((long long)fr * g.hop - g.center_pad) + 2 * odd;
        const int L = (int)g.length;
#pragma unroll 1
        for (int q = 0; q < E; ++q) {
            const int m = t + 64 * q;
            bool z0, z1;
            const int j0 = padded_index(s0 + 4 * m, L, g.pad_mode, &z0);
            const int j1 = padded_index(s0 + 4 * m + 1, L, g.pad_mode, &z1);
            const float a0 = rp[j0], a1 = rp[j1];
            xa[m] = mkc(z0 ? 0.0f : a0, z1 ? 0.0f : a1);
        }
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = xa[t + 64 * q];
        wave_lds_fence();
    };

    while (unit < end) {
        unsigned ask = 0;
        if (t == 0) ask = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const long long g0 = ((long long)row * T + fr) * (MEL ? mel.n_mels : LENF);     // this frame's row in the frame-major output
        const bool interior = mode == 1;
        cf alo[8], am[8], amid, blo[8], bm[8], bmid;
        {
            cf v[E];
            if (interior) {
#pragma unroll
                for (int q = 0; q < E; ++q) v[q] = mkc(raw[q].x, raw[q].y);
            } else {
                gather(v, 0);
            }
            transform(v, winA, alo, am, amid);
        }
        {
            cf v[E];
            if (interior) {
#pragma unroll
                for (int q = 0; q < E; ++q) v[q] = mkc(raw[q].z, raw[q].w);
            } else {
                gather(v, 1);
            }
            transform(v, winB, blo, bm, bmid);
        }
        const int nxt = (int)__builtin_amdgcn_readfirstlane(ask);
        // ---- combine + split (stft_n4096_kernel's formulas):  P = W_2048^k B[k], Q = conj(W_2048^k) B[1024-k];
        //   Z[k] = A[k] + P, Z[1024+k] = A[k] - P, Z[2048-k] = A[1024-k] + Q, Z[1024-k] = A[1024-k] - Q;
        //   bins (k, 2048-k) from (Z[k], Z[2048-k]) with W_4096^k; bins (1024-k, 1024+k) from (Z[1024-k], Z[1024+k]) with
        //   W_4096^(1024-k) = -i conj(W_4096^k).  k = 0 yields DC, Nyquist and bin 1024; the self-paired k = 512 is lane 0's extra.
        const int a = MEL ? 0 : (int)(g0 & 3);
        float* const stage = reinterpret_cast<float*>(xa) + a;          // LDS and global share their 16-byte phase
        {
            cf w2k[8], w4 = w4k;
            asm volatile("" : "+v"(w4));                                    // (keeps the eight W_4096^k products out of the loop's invariants: they would be spilled)
            const f4* pl = reinterpret_cast<const f4*>(ptwl) + t;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f4 x = pl[u * 64];
                w2k[2 * u] = mkc(x.x, x.y);
                w2k[2 * u + 1] = mkc(x.z, x.w);
            }
            cf w4t[8];
            if constexpr (TBL) {
                const f4* ql = reinterpret_cast<const f4*>(w4l) + t;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f4 x = ql[u * 64];
                    w4t[2 * u] = mkc(x.x, x.y);
                    w4t[2 * u + 1] = mkc(x.z, x.w);
                }
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int k = t + 64 * p;
                const cf pp = cmul(blo[p], w2k[p]);
                const cf qq = cmul_conj(bm[p], w2k[p]);
                const cf zk = cadd(alo[p], pp), zk2 = csub(alo[p], pp);     // Z[k], Z[1024 + k]
                const cf zp = cadd(am[p], qq), zm = csub(am[p], qq);        // Z[2048 - k], Z[1024 - k]
                const cf wq = TBL ? w4t[p] : mul_w64(w4, p);                // W_4096^k (without the table: W_4096^t W_64^p)
                cf x0, x1;                                                  // W_4096^(1024 - k) = -i conj(W_4096^k)
                r2c_power_pair_x2_mirror(zk, zp, zm, zk2, wq, x0, x1);      // (|X[k]|^2, |X[2048-k]|^2), (|X[1024-k]|^2, |X[1024+k]|^2)
                stage[k] = spectral_row_value<MODE>(x0.x, ep);
                stage[2048 - k] = spectral_row_value<MODE>(x0.y, ep);
                stage[1024 - k] = spectral_row_value<MODE>(x1.x, ep);
                stage[1024 + k] = spectral_row_value<MODE>(x1.y, ep);
            }
            if (t == 0) {                                                   // k = 512: bins 512 and 1536
                const cf pm = mul_neg_i(bmid);                              // W_2048^512 = -i
                const cf z5 = cadd(amid, pm), z15 = csub(amid, pm);
                const cf x5 = F::r2c_power_x2(z5, z15, mkc(TAC_SQRT_HALF, -TAC_SQRT_HALF));     // W_4096^512
                stage[512] = spectral_row_value<MODE>(x5.x, ep);
                stage[1536] = spectral_row_value<MODE>(x5.y, ep);
            }
            if (MEL && t < 3) stage[LENF + t] = 0.0f;                       // slack taps carry zero weights: keep them finite
            wave_lds_fence();
        }
        // ---- the next frame's samples are requested now, ahead of this row's stores (one in-order vmcnt)
        __builtin_amdgcn_sched_barrier(0);
        request(nxt);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MEL) {
            // ---- filterbank (reference functional.py:172-184) [+ dB, :291-296]: a step is one 16-byte weight read, one 16-byte row
            //      read and two packed FMAs; slot sl runs the step pairs of its longest piece, the next pair read before this one is used
            const bool fast_db = mel.amin >= 1.1754944e-38f;               // (uniform) hardware log2 unless the clamp admits denormals
            const float ten_log10_ref = 10.0f * mel.log10_ref;
            const f4* wp = reinterpret_cast<const f4*>(mwl) + t;
            float* const orow = mel.out + g0;
            // the step pairs of one slot from the lane's first bin: the next pair is read before this one is used
            auto run_slot = [&](int sl, int pairs) -> float {
                const f4* pp = reinterpret_cast<const f4*>(stage + mlo[64 * sl + t]);
                cf acc0 = mkc(0.0f, 0.0f), acc1 = mkc(0.0f, 0.0f);
                f4 w0 = wp[0], w1 = wp[64], p0 = pp[0], p1 = pp[1];
#pragma unroll 1
                for (int c = 1; c < pairs; ++c) {
                    wp += 128;
                    pp += 2;
                    const f4 nw0 = wp[0], nw1 = wp[64], np0 = pp[0], np1 = pp[1];
                    acc0 = __builtin_elementwise_fma(mkc(w0.x, w0.y), mkc(p0.x, p0.y), acc0);
                    acc1 = __builtin_elementwise_fma(mkc(w0.z, w0.w), mkc(p0.z, p0.w), acc1);
                    acc0 = __builtin_elementwise_fma(mkc(w1.x, w1.y), mkc(p1.x, p1.y), acc0);
                    acc1 = __builtin_elementwise_fma(mkc(w1.z, w1.w), mkc(p1.z, p1.w), acc1);
                    w0 = nw0; w1 = nw1; p0 = np0; p1 = np1;
                }
                wp += 128;
                acc0 = __builtin_elementwise_fma(mkc(w0.x, w0.y), mkc(p0.x, p0.y), acc0);
                acc1 = __builtin_elementwise_fma(mkc(w0.z, w0.w), mkc(p0.z, p0.w), acc1);
                acc0 = __builtin_elementwise_fma(mkc(w1.x, w1.y), mkc(p1.x, p1.y), acc0);
                acc1 = __builtin_elementwise_fma(mkc(w1.z, w1.w), mkc(p1.z, p1.w), acc1);
                return (acc0.x + acc0.y) + (acc1.x + acc1.y);
            };
            if (mel.np == 0) {                                              // direct: slot sl, lane t is band 64 sl + t
#pragma unroll 1
                for (int sl = 0; sl < mel.rounds; ++sl) {
                    float val = run_slot(sl, __builtin_amdgcn_readfirstlane(mlo[64 * N4M_SLOTS + sl]));
                    if (mel.db) val = fast_db ? amp_to_db_fast(val, mel.amin, ten_log10_ref) : amp_to_db(val, mel.amin, mel.log10_ref);
                    const int cell = 64 * sl + t, band = mel.rev ? mel.n_mels - 1 - cell : cell;
                    if (cell < mel.n_mels) __builtin_nontemporal_store(val, orow + band);
                }
                wave_lds_fence();                         // the next frame's first-pass writes follow these reads
                unit = nxt;
                continue;
            }
            const int* mix = mlo + N4M_DESC_HEAD + t;
            float part[N4M_SLOTS];
#pragma unroll
            for (int sl = 0; sl < N4M_SLOTS; ++sl) {
                const int pairs = __builtin_amdgcn_readfirstlane(mlo[64 * N4M_SLOTS + sl]);
                part[sl] = pairs > 0 ? run_slot(sl, pairs) : 0.0f;
            }
            // the row is dead: the cells take its place, every band collects its pieces
            wave_lds_fence();
            float* const cells = reinterpret_cast<float*>(xa);
#pragma unroll
            for (int sl = 0; sl < N4M_SLOTS; ++sl) cells[64 * sl + t] = part[sl];
            if (t == 0) cells[N4M_ZERO_CELL] = 0.0f;
            wave_lds_fence();
#pragma unroll 1
            for (int r = 0; r < mel.rounds; ++r) {
                float val = 0.0f;
#pragma unroll 1
                for (int i = 0; i < mel.np; i += 4) {                       // (np is a multiple of four: lists are padded with the zero cell)
                    const int i0 = mix[0], i1 = mix[64], i2 = mix[128], i3 = mix[192];
                    mix += 256;
                    const float c0 = cells[i0], c1 = cells[i1], c2 = cells[i2], c3 = cells[i3];
                    val += (c0 + c1) + (c2 + c3);
                }
                if (mel.db) val = fast_db ? amp_to_db_fast(val, mel.amin, ten_log10_ref) : amp_to_db(val, mel.amin, mel.log10_ref);
                const int band = 64 * r + t;
                if (band < mel.n_mels) __builtin_nontemporal_store(val, orow + band);
            }
            wave_lds_fence();                             // the next frame's first-pass writes follow these reads
            unit = nxt;
            continue;
        }
        // ---- the row leaves as 1 + NST + 1 unconditional nontemporal stores (lanes past the end repeat a neighbour)
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
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int c = (t + 64 * i) < last ? (t + 64 * i) : last;
                __builtin_nontemporal_store(s4[c], g4 + c);
            }
        }
        {
            const int r = LENF - npre - 4 * nchunks;
            const int rmax = (r > 1 ? r : 1) - 1;
            const int ti = LENF - 1 - (t < rmax ? t : rmax);
            gdst[ti] = stage[ti];
        }
        wave_lds_fence();                                 // the next frame's first-pass writes follow these reads
        unit = nxt;
    }
}

}  // namespace tac
