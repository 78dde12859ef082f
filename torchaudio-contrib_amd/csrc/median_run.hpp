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
