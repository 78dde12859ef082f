// stft_n4096.hip — fft_length = 4096 (SURVEY cfg-4): one frame per wave from TWO 1024-point complex FFTs.
//
// A 4096-sample real frame is 2048 complex points z[n] = x[2n] + i·x[2n+1].  With 32 elements per lane the
// single-wave 2048-point transform (stft_kernel<2048, 32>) keeps 64 data registers, cannot hoist its twiddles and
// cannot afford the software pipelining of stft_pipe_kernel.  Decimating in time once more keeps everything inside
// the register budget of the n_fft = 2048 kernel:
//     A = FFT_1024(z[2m]),  B = FFT_1024(z[2m+1])                       (same WaveFft<1024, 16> code, run twice)
//     Z[k]      = A[k] + W_2048^k · B[k]
//     Z[2048-k] = A[1024-k] + conj(W_2048^k) · B[1024-k]                 (0 <= k < 1024, indices mod 1024)
//     2·X[k] = ev + W_4096^k·(-i·d),  2·X[2048-k] = conj(ev - W_4096^k·(-i·d)),  ev/d = Z[k] ± conj(Z[2048-k])
//     X[1024] = conj(A[0] - B[0])
// i.e. the radix-2 combine is folded into the R2C split.  Per pair index k = t + 64·i the twiddles factor into one
// lane register times compile-time constants: W_2048^k = W_2048^t·W_32^i, W_4096^k = W_4096^t·W_64^i.
// One 16-byte load per lane fetches z[2m] and z[2m+1] together, so a frame is 16 dwordx4 requests; they are issued
// one frame ahead, before the previous frame's (nontemporal, unconditional) row stores — the stft_pipe_kernel
// recipe.  Two frame buffers per wave: one 8-wave workgroup per CU (157 KB of LDS with the shared window table; round 1's
// two 3-wave workgroups left two SIMDs with a single wave), frames drawn from a workgroup counter.
//
// Round 6: this kernel serves the COMPLEX rows; the real-valued rows (|X|, |X|^2, their dB forms) and the one-launch Melspectrogram
// chain run stft_n4096_s3_kernel (stft_n4096_s3.hpp: the two transforms one after the other through ONE area, twelve waves) — its
// launchers, the filterbank table builder of the fused chain (pack_n4096_mel) and the routing between the two are below.
#include "host_common.hpp"
#include "lane_placement.hpp"

#include <algorithm>
#include <vector>

namespace tac {

constexpr int N4K_WAVES = 8;
typedef float f4 __attribute__((ext_vector_type(4)));

// multiply by W_64^i = exp(-2*pi*i*i/64), 0 <= i < 16 (compile-time after unrolling)
__device__ __forceinline__ cf mul_w64(cf v, int i) {
    constexpr float C[16] = {1.0000000000e+00f, 9.9518472667e-01f, 9.8078528040e-01f, 9.5694033573e-01f, 9.2387953251e-01f, 8.8192126435e-01f, 8.3146961230e-01f, 7.7301045336e-01f, 7.0710678119e-01f, 6.3439328416e-01f, 5.5557023302e-01f, 4.7139673683e-01f, 3.8268343237e-01f, 2.9028467725e-01f, 1.9509032202e-01f, 9.8017140330e-02f};
    constexpr float S[16] = {-0.0000000000e+00f, -9.8017140330e-02f, -1.9509032202e-01f, -2.9028467725e-01f, -3.8268343237e-01f, -4.7139673683e-01f, -5.5557023302e-01f, -6.3439328416e-01f, -7.0710678119e-01f, -7.7301045336e-01f, -8.3146961230e-01f, -8.8192126435e-01f, -9.2387953251e-01f, -9.5694033573e-01f, -9.8078528040e-01f, -9.9518472667e-01f};
    if (i == 0) return v;
    return cmulc(v, C[i & 15], S[i & 15]);
}

// a · conj(w): (a.x·w.x + a.y·w.y, a.y·w.x - a.x·w.y)
__device__ __forceinline__ cf cmul_conj(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                 // a.x·(w.x, -w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));   // + a.y·(w.y, w.x)
    return r;
}

template <int MODE>
__global__ void __launch_bounds__(N4K_WAVES * 64, 2)
stft_n4096_kernel(FrameGeom g, Tables tb1k, Tables tb4k, StftEpilogue ep) {
    using F = WaveFft<1024, 16>;
    constexpr int E = 16, NCH = 1024, NBINS = 2049;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    const int t = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WS = ((F::PADDED + 1) / 2) * 2;
    cf* const bufA = smem + w * 2 * WS;
    cf* const bufB = bufA + WS;
    // window: for lane t and q < 16 the four window values of samples 4m .. 4m+3, m = t + 64q; 272-byte rows
    constexpr int WROW = E + 1;
    f4* const wl4 = reinterpret_cast<f4*>(smem + N4K_WAVES * 2 * WS);
    for (int m = threadIdx.x; m < NCH; m += N4K_WAVES * 64) {
        const cf wa = window_pair(g, 2 * m), wb = window_pair(g, 2 * m + 1);
        wl4[(m & 63) * WROW + (m >> 6)] = f4{wa.x, wa.y, wb.x, wb.y};
    }
    cf tw[F::NTW];
    F::load_twiddles(tw, tb1k.w_nc, t);
    const cf w2k = tb4k.w_nc[t];                          // W_2048^t
    const cf w4k = tb4k.w_n[t];                           // W_4096^t

    const int T = (int)g.n_frames;
    const int total = (int)g.rows * T;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * chunk;
    const int end = begin + chunk < total ? begin + chunk : total;
    constexpr int LENF = (MODE == 0 ? 2 : 1) * NBINS;
    constexpr int NST = ((LENF >> 2) + 63) / 64;          // 16-byte wave-stores per output row: 17 (complex) / 9
    const float hscale = 0.5f * g.scale;

    // one frame ahead: 16 requests of 16 bytes per lane (samples 4m .. 4m+3, m = t + 64q)
    f4 raw[E];
    auto prefetch = [&](int unit) -> bool {
        const int urow = unit / T, uframe = unit - urow * T;
        const long long start = (long long)uframe * g.hop - g.center_pad;
        const bool ok = g.vec4_ok && start >= 0 && start + 4096 <= g.length;
        if (ok) {
            const f4* src = reinterpret_cast<const f4*>(g.wave + (long long)urow * g.row_stride + start);
#pragma unroll
            for (int q = 0; q < E; ++q) raw[q] = src[t + 64 * q];
        }
        return ok;
    };
    // frames are taken from a workgroup counter, not dealt out in fixed strides (the older of the two waves of a SIMD
    // wins the issue arbitration and would finish a fixed share long before the other)
    unsigned* const next_unit = reinterpret_cast<unsigned*>(wl4 + 64 * WROW);
    if (threadIdx.x == 0) *next_unit = (unsigned)(begin + N4K_WAVES);
    auto grab = [&]() -> int {
        unsigned v = 0;
        if (t == 0) v = __hip_atomic_fetch_add(next_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        const int urow = unit / T;
        const int uframe = unit - urow * T;
        cf va[1][E], vb[1][E];
        cf* const la[1] = {bufA};
        cf* const lb[1] = {bufB};
        if (pre) {
            const f4* wp = wl4 + t * WROW;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const f4 wv = wp[q];
                va[0][q] = cmul_elem(mkc(raw[q].x, raw[q].y), mkc(wv.x, wv.y));
                vb[0][q] = cmul_elem(mkc(raw[q].z, raw[q].w), mkc(wv.z, wv.w));
            }
        } else {
            // frames touching the padding: gathered sample by sample through the frame buffers (rolled loop)
            const float* rp = g.wave + (long long)urow * g.row_stride;
            const int s0 = (int)((long long)uframe * g.hop - g.center_pad);
            const int L = (int)g.length;
#pragma unroll 1
            for (int q = 0; q < E; ++q) {
                const int m = t + 64 * q;
                float smp[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bool zero;
                    const int j = padded_index(s0 + 4 * m + c, L, g.pad_mode, &zero);
                    const float vsmp = rp[j];
                    smp[c] = zero ? 0.0f : vsmp;
                }
                const cf wa = window_pair(g, 2 * m), wb = window_pair(g, 2 * m + 1);
                bufA[lds_pad(m)] = mkc(smp[0] * wa.x, smp[1] * wa.y);
                bufB[lds_pad(m)] = mkc(smp[2] * wb.x, smp[3] * wb.y);
            }
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < E; ++q) {
                va[0][q] = bufA[lds_pad(t + 64 * q)];
                vb[0][q] = bufB[lds_pad(t + 64 * q)];
            }
            wave_lds_fence();
        }
        // both transforms keep the lower half of their spectrum in registers (the HALF form of fft_core.hpp): a lane then owns the
        // eight PAIRS (k, 1024 - k), k = t + 64 p, and needs only the partners A[1024 - k], B[1024 - k] from LDS
        F::template run<1, NoStamp, true>(va, la, tw, t, st, t);
        F::template run<1, NoStamp, true>(vb, lb, tw, t, st, t);

        // request the next frame now: it lands while this frame is combined, split, staged and stored
        __builtin_amdgcn_sched_barrier(0);
        {
            pre = false;
            if (nxt < end) pre = prefetch(nxt);
        }
        __builtin_amdgcn_sched_barrier(0);

        const long long g0 = ((long long)urow * T + uframe) * LENF;
        const int a = (int)(g0 & 3);
        float* const stage = reinterpret_cast<float*>(bufA) + a;     // LDS and global share their 16-byte phase
        {
            // round 3: each pair is combined ONCE into its four bins (it used to be formed twice, once from either end, at four
            // LDS reads per bin pair):  P = W_2048^k B[k], Q = conj(W_2048^k) B[1024-k];
            //   Z[k] = A[k] + P, Z[1024+k] = A[k] - P, Z[2048-k] = A[1024-k] + Q, Z[1024-k] = A[1024-k] - Q;
            //   bins (k, 2048-k) from (Z[k], Z[2048-k]) with W_4096^k; bins (1024-k, 1024+k) from (Z[1024-k], Z[1024+k]) with
            //   W_4096^(1024-k) = -i conj(W_4096^k).  k = 0 yields DC, Nyquist and bin 1024 by the same formulas; the self-paired
            //   k = 512 is lane 0's extra.
            cf am[8], bm[8];
            {
                const cf* const pa = bufA + lds_pad(NCH - t);
                const cf* const pb = bufB + lds_pad(NCH - t);
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const cf za = pa[-lds_pad_c(p * 64)], zb = pb[-lds_pad_c(p * 64)];
                    am[p] = (p == 0 && t == 0) ? va[0][F::reg_of_spectrum(0)] : za;
                    bm[p] = (p == 0 && t == 0) ? vb[0][F::reg_of_spectrum(0)] : zb;
                }
            }
            const cf amid = bufA[lds_pad(NCH / 2)], bmid = bufB[lds_pad(NCH / 2)];
            cf x0[8], x1[8], y0[8], y1[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const cf ak = va[0][F::reg_of_spectrum(p)], bk = vb[0][F::reg_of_spectrum(p)];
                const cf pp = cmul(mul_w32(bk, p), w2k);
                const cf qq = cmul_conj(mul_w32(bm[p], 32 - p), w2k);
                const cf zk = cadd(ak, pp), zk2 = csub(ak, pp);         // Z[k], Z[1024 + k]
                const cf zp = cadd(am[p], qq), zm = csub(am[p], qq);    // Z[2048 - k], Z[1024 - k]
                const cf wq = mul_w64(w4k, p);                          // W_4096^k = W_4096^t W_64^p
                const cf wr = mkc(-wq.y, -wq.x);                        // W_4096^(1024 - k)
                if constexpr (MODE != 0) {
                    x0[p] = cscale(F::r2c_power_x2(zk, zp, wq), hscale * hscale);     // (|X[k]|^2, |X[2048 - k]|^2)
                    x1[p] = cscale(F::r2c_power_x2(zm, zk2, wr), hscale * hscale);    // (|X[1024 - k]|^2, |X[1024 + k]|^2)
                } else {
                    F::r2c_split_x2(zk, zp, wq, x0[p], y0[p]);
                    F::r2c_split_x2(zm, zk2, wr, x1[p], y1[p]);
                    x0[p] = cscale(x0[p], hscale); y0[p] = cscale(y0[p], hscale);
                    x1[p] = cscale(x1[p], hscale); y1[p] = cscale(y1[p], hscale);
                }
            }
            cf x5 = mkc(0.0f, 0.0f), y5 = mkc(0.0f, 0.0f);              // k = 512: bins 512 and 1536
            {
                const cf wi = mkc(0.0f, -1.0f);                         // W_2048^512
                const cf pm = cmul(bmid, wi);
                const cf z5 = cadd(amid, pm), z15 = csub(amid, pm);
                const cf w5 = mkc(0.70710678118654752f, -0.70710678118654752f);     // W_4096^512
                if constexpr (MODE != 0) x5 = cscale(F::r2c_power_x2(z5, z15, w5), hscale * hscale);
                else {
                    F::r2c_split_x2(z5, z15, w5, x5, y5);
                    x5 = cscale(x5, hscale); y5 = cscale(y5, hscale);
                }
            }
            wave_lds_fence();                                         // every A, B of this frame is in registers
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int k = t + 64 * p;
                if constexpr (MODE == 0) {
                    cf* const sc = reinterpret_cast<cf*>(stage);
                    sc[k] = x0[p]; sc[2048 - k] = y0[p];
                    sc[1024 - k] = x1[p]; sc[1024 + k] = y1[p];
                } else {
                    stage[k] = spectral_row_value<MODE>(x0[p].x, ep);
                    stage[2048 - k] = spectral_row_value<MODE>(x0[p].y, ep);
                    stage[1024 - k] = spectral_row_value<MODE>(x1[p].x, ep);
                    stage[1024 + k] = spectral_row_value<MODE>(x1[p].y, ep);
                }
            }
            if (t == 0) {
                if constexpr (MODE == 0) {
                    reinterpret_cast<cf*>(stage)[512] = x5;
                    reinterpret_cast<cf*>(stage)[1536] = y5;
                } else {
                    stage[512] = spectral_row_value<MODE>(x5.x, ep);
                    stage[1536] = spectral_row_value<MODE>(x5.y, ep);
                }
            }
            wave_lds_fence();
        }
        // the row leaves as 1 + NST + 1 unconditional nontemporal stores (lanes past the end repeat a neighbour)
        float* const gdst = ep.out + g0;
        const int npre = (4 - a) & 3;
        const int nchunks = (LENF - npre) >> 2;
        {
            const int hmax = (npre > 1 ? npre : 1) - 1;
            const int hi = t < hmax ? t : hmax;
            gdst[hi] = stage[hi];
        }
        const f4* const s4 = reinterpret_cast<const f4*>(stage + npre);
        f4* const g4 = reinterpret_cast<f4*>(gdst + npre);
        const int last = nchunks - 1;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = (t + 64 * i) < last ? (t + 64 * i) : last;
            __builtin_nontemporal_store(s4[c], g4 + c);
        }
        {
            const int r = LENF - npre - 4 * nchunks;
            const int rmax = (r > 1 ? r : 1) - 1;
            const int ti = LENF - 1 - (t < rmax ? t : rmax);
            gdst[ti] = stage[ti];
        }
        wave_lds_fence();   // next iteration's first-pass writes must follow these reads
        unit = nxt;
    }
}

}  // namespace tac

#include "stft_n4096_s3.hpp"

namespace tac {

template <int MODE, int WAVES, bool MEL, bool TBL = !MEL>
static int launch_n4096_s3(const FrameGeom& g, const Tables& tb2k, const Tables& tb4k, const StftEpilogue& ep, const N4Mel& mel,
                           hipStream_t stream) {
    const long long units = g.rows * g.n_frames;
    if (units >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    const size_t bytes = n4096_s3_lds_bytes(WAVES) + (MEL ? n4096_mel_lds_bytes(mel.rounds, mel.np, mel.wtot) : 0) + (TBL ? N4S_ROW_TABLE_BYTES : 0);
    if (bytes > 160 * 1024) return TAC_E_UNSUPPORTED;
    long long blocks = (units + WAVES - 1) / WAVES;
    const long long cap = (long long)device_cu_count();      // one workgroup per CU
    if (blocks > cap) blocks = cap;
    auto kern = stft_n4096_s3_kernel<MODE, WAVES, MEL, TBL>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), bytes, stream, g, tb2k, tb4k, ep, mel);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// ---- the filterbank table of the fused form (tac_melbank_pack, n_fft == 4096; layout: stft_n4096_s3.hpp).  A band's run of quads
// (from its first non-zero bin rounded down to a multiple of four) is cut into ceil(quads / C) pieces of about equal length; all pieces,
// longest first, fill the cells row by row; slot s runs the (even) number of steps of its longest piece.  C is the cap that minimises
// the model's LDS cycles per frame (two 16-byte reads per step; cells + gather cost what ~35 steps do — measured in one process: the
// 54-step uncut layout of a 96-band bank on eight waves 0.1118 ms, its 26-step cut layout on twelve 0.1161; 128 bands: 40 uncut steps
// 0.1084, 22 cut 0.1124) among those that need at most N4M_SLOTS slots and N4M_MAX_PIECES pieces per band and fit the LDS beside eight
// waves: bands are cut only where the uncut table does not fit beside eight waves (banks of ~40 bands, 300 bins wide).  A piece's first bin is then moved down within its slot's slack by the bank-aware matching
// (lane_placement.hpp), and where its padded run would leave the row and its three zeroed slack floats.
// info: [0] floats of weights, [1] slots in use, [2] N4M_MARK, [3] steps, [4] waves per workgroup the table leaves room for,
// [5] pieces per band in the mix table (0: uncut), [6] rounds of 64 bands, [7] uncut cells in reversed band order.
// to_host: wpack / desc are HOST buffers (tac_melbank_pack_host: the same tables without a device, for the CPU tests)
int pack_n4096_mel(const std::vector<float>& h, int n_freqs, int n_mels, float* wpack, int wpack_cap, int32_t* desc, int desc_cap,
                   int32_t* info_host, hipStream_t stream, bool to_host) {
    constexpr int LIMIT = 2049 + 3;
    if (n_freqs != 2049 || n_mels < 1 || n_mels > N4M_MAX_MELS) return TAC_E_UNSUPPORTED;
    const int rounds = (n_mels + 63) / 64;
    std::vector<int> lo(n_mels, 0), hi(n_mels, 0), first(n_mels, 0), quads(n_mels, 0);
    int widest = 1;
    for (int m = 0; m < n_mels; ++m) {
        int l0 = n_freqs, h0 = 0;
        for (int f = 0; f < n_freqs; ++f)
            if (h[(size_t)f * n_mels + m] != 0.0f) { l0 = f < l0 ? f : l0; h0 = f + 1; }
        if (h0 > l0) {
            lo[m] = l0;
            hi[m] = h0;
            first[m] = l0 & ~3;
            quads[m] = (h0 - first[m] + 3) / 4;
            widest = std::max(widest, quads[m]);
        }
    }
    struct Piece { int band, start, q; };
    const bool rev = quads[n_mels - 1] > quads[0];                          // uncut cells from the widest end of the bank
    auto cut = [&](int cap, std::vector<Piece>& pieces, int& np_max) {
        pieces.clear();
        np_max = 1;
        for (int mm = 0; mm < n_mels; ++mm) {
            const int m = rev ? n_mels - 1 - mm : mm;
            if (!quads[m]) {                                                // a band without support still owns a cell (it sums to zero)
                pieces.push_back({m, 0, 0});
                continue;
            }
            const int np = (quads[m] + cap - 1) / cap, base = quads[m] / np, extra = quads[m] % np;
            np_max = std::max(np_max, np);
            for (int i = 0, at = first[m]; i < np; ++i) {
                const int q = base + (i < extra ? 1 : 0);
                pieces.push_back({m, at, q});
                at += 4 * q;
            }
        }
        if (np_max > 1) std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& x, const Piece& y) { return x.q > y.q; });
    };                                                                      // (uncut: cell 64 r + l stays band 64 r + l)
    std::vector<Piece> pieces, best;
    long long best_cost = -1;
    int best_np = 1;
    for (int cap = 1; cap <= widest; ++cap) {
        int np = 1;
        cut(cap, pieces, np);
        const int slots = ((int)pieces.size() + 63) / 64;
        if (np > N4M_MAX_PIECES || slots > N4M_SLOTS) continue;
        long long steps = 0;
        for (int s = 0; s < slots; ++s) {
            int q = 0;
            for (int c = 64 * s; c < 64 * s + 64 && c < (int)pieces.size(); ++c) q = std::max(q, pieces[c].q);
            steps += std::max(2, (q + 1) & ~1);
        }
        const int npad = np > 1 ? (np + 3) & ~3 : 0;
        if (n4096_s3_lds_bytes(8) + n4096_mel_lds_bytes(rounds, npad, (int)(256 * steps)) > 160 * 1024) continue;
        const long long cost = 8 * steps + (np > 1 ? 300 + 2LL * rounds * npad : 0) + 16 * slots;
        if (best_cost < 0 || cost <= best_cost) {
            best_cost = cost;
            best = pieces;
            best_np = np;
        }
    }
    if (best_cost < 0) return TAC_E_UNSUPPORTED;                            // bands too many or too wide for the LDS: the two-launch chain
    const int ncell = (int)best.size(), slots = std::max(1, (ncell + 63) / 64), np = best_np > 1 ? (best_np + 3) & ~3 : 0;     // (mix lists in fours; 0: direct)
    int slot_steps[N4M_SLOTS] = {0, 0, 0, 0, 0, 0}, steps = 0;
    for (int s = 0; s < slots; ++s) {
        int q = 0;
        for (int c = 64 * s; c < 64 * s + 64 && c < ncell; ++c) q = std::max(q, best[c].q);
        slot_steps[s] = std::max(2, (q + 1) & ~1);
        if (4 * slot_steps[s] > LIMIT) return TAC_E_UNSUPPORTED;
        steps += slot_steps[s];
    }
    const long long wtot = 256LL * steps;
    int waves = 0;
    // twelve or eleven waves with the twiddle tables of the row-store form beside the bank (stft_n4096_s3.hpp, TBL: eleven waves with them beat
    // twelve without by 5 - 9 %), else twelve / eleven / eight without asking (eight always have the room; ten waves — 3, 3, 2, 2 per SIMD —
    // measured slower than eight): profiles/r06/ab/batch25_mel4096_tables.txt
    for (int pass = 0; pass < 2 && !waves; ++pass)
        for (int wv : {12, 11, 8})
            if (!waves && (pass == 1 || wv > 8) &&
                n4096_s3_lds_bytes(wv) + n4096_mel_lds_bytes(rounds, np, (int)wtot) + (pass == 0 ? N4S_ROW_TABLE_BYTES : 0) <= 160 * 1024)
                waves = wv;
    const int mix_rows = n4096_mix_rows(rounds, np);
    if (!waves || wtot > wpack_cap || N4M_DESC_HEAD + 64 * mix_rows > desc_cap) return TAC_E_UNSUPPORTED;   // the two-launch chain
    // bank-aware first bins per slot: a cell's run may start up to (slot steps - its quads) quads earlier
    std::vector<int> cstart(64 * slots, 0), clen(64 * slots, 0);
    for (int c = 0; c < ncell; ++c) {
        cstart[c] = best[c].start;
        clen[c] = 4 * best[c].q;
    }
    const std::vector<int> placed = place_band_starts(slots, slot_steps, cstart, clen);
    std::vector<float> wp((size_t)wtot, 0.0f);
    std::vector<int32_t> dd((size_t)N4M_DESC_HEAD + 64 * mix_rows, 0);
    for (int i = 0; i < 64 * mix_rows; ++i) dd[(size_t)N4M_DESC_HEAD + i] = N4M_ZERO_CELL;
    std::vector<int> filled(n_mels, 0);
    int base = 0;
    for (int s = 0; s < slots; ++s) {
        for (int l = 0; l < 64; ++l) {
            const int c = 64 * s + l;
            int f0 = c < ncell ? placed[c] : 0;
            if (f0 + 4 * slot_steps[s] > LIMIT) f0 = (LIMIT - 4 * slot_steps[s]) & ~3;
            dd[c] = f0;
            if (c >= ncell) continue;
            const Piece& pc = best[c];
            for (int j = 0; j < slot_steps[s]; ++j)
                for (int u = 0; u < 4; ++u) {
                    const int bin = f0 + 4 * j + u;
                    const bool live = bin >= pc.start && bin < pc.start + 4 * pc.q && bin >= lo[pc.band] && bin < hi[pc.band];
                    wp[((size_t)(base + j) * 64 + l) * 4 + u] = live ? h[(size_t)bin * n_mels + pc.band] : 0.0f;
                }
            if (np) dd[(size_t)N4M_DESC_HEAD + ((size_t)(pc.band / 64) * np + filled[pc.band]++) * 64 + (pc.band & 63)] = c;

        }
        dd[(size_t)64 * N4M_SLOTS + s] = slot_steps[s] / 2;
        base += slot_steps[s];
    }
    if (to_host) {
        std::copy(wp.begin(), wp.end(), wpack);
        std::copy(dd.begin(), dd.end(), desc);
    } else {
        TAC_HIP(hipMemcpyAsync(wpack, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        TAC_HIP(hipMemcpyAsync(desc, dd.data(), dd.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        TAC_HIP(hipStreamSynchronize(stream));
    }
    info_host[0] = (int32_t)wtot;
    info_host[1] = slots;
    info_host[2] = N4M_MARK;
    info_host[3] = steps;
    info_host[4] = waves;
    info_host[5] = np;
    info_host[6] = rounds;
    info_host[7] = (np == 0 && rev) ? 1 : 0;
    return TAC_OK;
}

// Melspectrogram (-> AmplitudeToDb) at fft_length 4096 in one launch (tac_melspec_sparse_f32): TAC_E_UNSUPPORTED for the geometries
// the twelve-wave form declines (two-sided output, frames that are not 16-byte aligned, rows shorter than one frame)
int launch_n4096_mel(const FrameGeom& g, float power, const float* wpack, const int32_t* desc, const int32_t* info_host, int n_mels,
                     int db, float amin, float log10_ref, float* out, hipStream_t stream) {
    if (info_host[2] != N4M_MARK || info_host[1] < 1 || info_host[1] > N4M_SLOTS || info_host[0] != 256 * info_host[3] ||
        info_host[5] < 0 || info_host[5] > N4M_MAX_PIECES || (info_host[5] & 3) || info_host[6] != (n_mels + 63) / 64 || n_mels > N4M_MAX_MELS)
        return TAC_E_INVALID;
    if (!g.vec4_ok || g.length < 4096 || (power != 1.0f && power != 2.0f)) return TAC_E_UNSUPPORTED;
    Tables tb2k, tb4k;
    int rc = get_tables(2048, &tb2k);
    if (rc != TAC_OK) return rc;
    rc = get_tables(4096, &tb4k);
    if (rc != TAC_OK) return rc;
    const StftEpilogue ep{nullptr, 1, 1, power, 0, 0.0f, 0.0f};
    const N4Mel mel{wpack, desc, info_host[5], info_host[6], info_host[0], n_mels, db, info_host[7], amin, log10_ref, out};
    const bool p2 = power == 2.0f;
    const int waves = info_host[4];
    // the twiddle tables of the row-store form where the bank leaves them the room
    const bool tbl = n4096_s3_lds_bytes(waves) + n4096_mel_lds_bytes(mel.rounds, mel.np, mel.wtot) + N4S_ROW_TABLE_BYTES <= 160 * 1024;
#define TAC_N4M_CASE(W)                                                                                                     \
    case W:                                                                                                                \
        if (tbl) return p2 ? launch_n4096_s3<1, W, true, true>(g, tb2k, tb4k, ep, mel, stream) : launch_n4096_s3<2, W, true, true>(g, tb2k, tb4k, ep, mel, stream); \
        return p2 ? launch_n4096_s3<1, W, true, false>(g, tb2k, tb4k, ep, mel, stream) : launch_n4096_s3<2, W, true, false>(g, tb2k, tb4k, ep, mel, stream);
    switch (waves) {
        TAC_N4M_CASE(12) TAC_N4M_CASE(11) TAC_N4M_CASE(8)
#undef TAC_N4M_CASE
        default: return TAC_E_INVALID;
    }
}

template <int MODE>
static int launch_n4096(const FrameGeom& g, const Tables& tb1k, const Tables& tb4k, const StftEpilogue& ep,
                        hipStream_t stream) {
    using F = WaveFft<1024, 16>;
    const long long units = g.rows * g.n_frames;
    if (units >= 0x7fffffffLL) return TAC_E_UNSUPPORTED;
    constexpr int WS = ((F::PADDED + 1) / 2) * 2;
    const size_t bytes = (size_t)N4K_WAVES * 2 * WS * sizeof(cf) + (size_t)64 * 17 * sizeof(f4) + 16;
    long long blocks = (units + N4K_WAVES - 1) / N4K_WAVES;
    const long long cap = (long long)device_cu_count();      // one 8-wave workgroup per CU (157 KB of LDS)
    if (blocks > cap) blocks = cap;
    auto kern = stft_n4096_kernel<MODE>;
    TAC_HIP(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(N4K_WAVES * 64), bytes, stream, g, tb1k, tb4k, ep);
    TAC_HIP(hipGetLastError());
    return TAC_OK;
}

// Entry used by stft_kernels.hip's dispatcher: returns TAC_E_UNSUPPORTED when this form does not apply (two-sided
// output, |X|^p with p outside {1, 2}, frames that are not 16-byte aligned) so that the generic kernel takes over.
int try_launch_n4096(const FrameGeom& g, const StftEpilogue& ep, int mode, hipStream_t stream) {
    if (!ep.onesided || !g.vec4_ok) return TAC_E_UNSUPPORTED;
    int pmode = -1;
    if (mode == 0) pmode = 0;
    else if (ep.power == 2.0f) pmode = ep.db ? 3 : 1;
    else if (ep.power == 1.0f) pmode = ep.db ? 4 : 2;
    if (pmode < 0) return TAC_E_UNSUPPORTED;
    Tables tb1k, tb4k;
    int rc = get_tables(2048, &tb1k);
    if (rc != TAC_OK) return rc;
    rc = get_tables(4096, &tb4k);
    if (rc != TAC_OK) return rc;
    if (pmode >= 1 && g.length >= 4096) {       // real-valued rows: the twelve-wave form (stft_n4096_s3.hpp)
        const N4Mel none{};
        switch (pmode) {
            case 1: return launch_n4096_s3<1, 12, false>(g, tb1k, tb4k, ep, none, stream);
            case 2: return launch_n4096_s3<2, 12, false>(g, tb1k, tb4k, ep, none, stream);
            case 3: return launch_n4096_s3<3, 12, false>(g, tb1k, tb4k, ep, none, stream);
            default: return launch_n4096_s3<4, 12, false>(g, tb1k, tb4k, ep, none, stream);
        }
    }
    switch (pmode) {
        case 0: return launch_n4096<0>(g, tb1k, tb4k, ep, stream);
        case 1: return launch_n4096<1>(g, tb1k, tb4k, ep, stream);
        case 2: return launch_n4096<2>(g, tb1k, tb4k, ep, stream);
        case 3: return launch_n4096<3>(g, tb1k, tb4k, ep, stream);
        default: return launch_n4096<4>(g, tb1k, tb4k, ep, stream);
    }
}

}  // namespace tac
