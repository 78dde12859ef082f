// median_run.hpp — register sorting networks of the HPSS median filters (hpss.hip).  Plain C++ on purpose: the same
// header compiles for the host, where tests/test_host_api.py::test_median_run_matches_nth_element checks the
// shared-sort selection against std::nth_element for every supported width.
#pragma once
#include <math.h>
#ifndef TAC_HD
#ifdef __HIPCC__
#define TAC_HD __host__ __device__ __forceinline__
#else
#define TAC_HD inline
#endif
#endif
TAC_HD void cswap_f(float& a, float& b) {
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}
// three values in order: on the device one instruction each (v_min3 / v_med3 / v_max3), on the host three comparators
TAC_HD void sort3_f(float& a, float& b, float& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float lo = __builtin_fminf(__builtin_fminf(a, b), c), md = __builtin_amdgcn_fmed3f(a, b, c),
                hi = __builtin_fmaxf(__builtin_fmaxf(a, b), c);
    a = lo;
    b = md;
    c = hi;
#else
    cswap_f(a, b);
    cswap_f(b, c);
    cswap_f(a, b);
#endif
}
// Batcher odd-even mergesort of N values held in registers (comparators that would touch an index >= N are the ones
// a +inf padding would make no-ops, so they are simply left out)
template <int N>
TAC_HD void sort_net(float (&a)[N]) {
#pragma unroll
    for (int p = 1; p < N; p *= 2)
#pragma unroll
        for (int k = p; k >= 1; k /= 2)
#pragma unroll
            for (int j = k % p; j + k < N; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < N && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cswap_f(a[i + j], a[i + j + k]);
}
// medians of the FOUR windows w[j .. j+K-1], j = 0..3, of K + 3 consecutive taps (K odd, >= 9).  The K - 3 taps all four
// windows share are sorted once; a window's median (rank m = (K-1)/2 of its K taps) is then the 4th smallest of
// {C[m-3..m]} and its three own taps clamped from below at C[m-4] — the rank-m element of the union cannot lie below
// C[m-3] (at most 3 own taps can precede it) nor above C[m].
template <int K>
TAC_HD void median_run4(const float (&w)[K + 3], float (&med)[4]) {
    static_assert(K >= 9 && (K & 1), "shared-sort form needs an odd K >= 9");
    constexpr int NCOM = K - 3, M = (K - 1) / 2;
    float c[NCOM];
#pragma unroll
    for (int i = 0; i < NCOM; ++i) c[i] = w[3 + i];
    sort_net<NCOM>(c);
    const float lo = c[M - 4], c0 = c[M - 3], c1 = c[M - 2], c2 = c[M - 1], c3 = c[M];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float e[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) e[u] = fmaxf(u < 3 - j ? w[j + u] : w[K + (u - (3 - j))], lo);
        cswap_f(e[0], e[1]);
        cswap_f(e[1], e[2]);
        cswap_f(e[0], e[1]);
        med[j] = fminf(fminf(fmaxf(c0, e[2]), fmaxf(c1, e[1])), fminf(fmaxf(c2, e[0]), c3));
    }
}

// medians of the EIGHT windows w[j .. j+K-1], j = 0..7, of K + 7 consecutive taps (K odd, >= 9) — two levels of sharing.
// Level 0: the K - 7 taps all eight windows share, D = sorted w[7 .. K-1], once.  Level 1: windows 0..3 share D and
// G = w[3..6], windows 4..7 share D and G = w[K .. K+3]: of the merged list C = D u G (what median_run4 sorts from scratch)
// only the ranks M-3 .. M are needed, M = (K-1)/2, and rank r of the union of two sorted lists is
//     min over t = 0..4 of max(D[r - t], G[t - 1])            (t = how many of its r + 1 smallest come from G),
// eight min / max per rank.  Level 2 as in median_run4: the window's median is the 4th smallest of those four ranks and its
// three own taps.  K = 31: 132 (sort 24) + 2 x (5 + 16) + 8 x 6.5 comparators = 56 min / max operations per median (run4: 97).
template <int K>
TAC_HD void median_run8(const float (&w)[K + 7], float (&med)[8]) {
    static_assert(K >= 9 && (K & 1), "shared-sort form needs an odd K >= 9");
    constexpr int N0 = K - 7, M = (K - 1) / 2;
    float d[N0];
#pragma unroll
    for (int i = 0; i < N0; ++i) d[i] = w[7 + i];
    sort_net<N0>(d);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = h == 0 ? w[3 + i] : w[K + i];
        sort_net<4>(g);
        float c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = M - 3 + q;
            float val = 0.0f;
            bool have = false;
#pragma unroll
            for (int t = 0; t <= 4; ++t) {
                const int idx = r - t;                         // r + 1 - t values come from D: its element idx is the largest
                if (idx > N0 - 1 || idx < -1) continue;        // (D has no such element / more than r + 1 taken from G)
                const int ic = idx < 0 ? 0 : idx;
                const float term = t == 0 ? d[ic] : (idx == -1 ? g[t - 1] : fmaxf(d[ic], g[t - 1]));
                val = have ? fminf(val, term) : term;
                have = true;
            }
            c[q] = val;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) e[u] = u < 3 - j ? w[4 * h + j + u] : w[4 * h + K + (u - (3 - j))];
            sort3_f(e[0], e[1], e[2]);
            med[4 * h + j] = fminf(fminf(fmaxf(c[0], e[2]), fmaxf(c[1], e[1])), fminf(fmaxf(c[2], e[0]), c[3]));
        }
    }
}

// any odd K >= 1: the shared-sort form from 9 up, below that a sort of its own per window (at most 16 comparators)
template <int K>
TAC_HD void median_run8_any(const float (&w)[K + 7], float (&med)[8]) {
    static_assert(K >= 1 && (K & 1), "odd widths only");
    if constexpr (K >= 9) {
        median_run8<K>(w, med);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v[K];
#pragma unroll
            for (int u = 0; u < K; ++u) v[u] = w[j + u];
            sort_net<K>(v);
            med[j] = v[K / 2];
        }
    }
}
