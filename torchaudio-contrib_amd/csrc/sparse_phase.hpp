// sparse_phase.hpp — the band-sparse filterbank contraction shared by melspec_sparse.hip and melspec_stream.hpp.
#pragma once
#include "mel_common.hpp"

namespace tac {

// phase B: one private dot product per (frame, band) — this thread's lane group owns the band list at dg, its lane
// within the group owns the frame whose power row starts at prow; results go to the frame's row of the output tile.
__device__ __forceinline__ void sparse_phase_b(const int* dg, const float* prow, const float* wlds, float* orow) {
    const int nb = dg[0];
    int4 dnext = *reinterpret_cast<const int4*>(dg + 4);                      // band, first bin, n8, weight offset
    for (int b = 0; b < nb; ++b) {
        const int4 d = dnext;
        dnext = *reinterpret_cast<const int4*>(dg + 4 + 4 * (b + 1 < nb ? b + 1 : b));   // next band's descriptor in flight
        const float* p = prow + d.y;
        const float4* w4 = reinterpret_cast<const float4*>(wlds + d.w);
        cf acc0 = mkc(0.0f, 0.0f), acc1 = mkc(0.0f, 0.0f);
        // 8 taps per trip: 2 weight vectors (LDS broadcast) + 4 eight-byte row reads (bands start on even bins,
        // tac_melbank_pack) feed 4 packed FMAs.  The loop is LDS-latency-bound at 2 waves/SIMD, so it is software
        // pipelined: trip j+1's six reads are issued before trip j's FMAs (the last trip re-reads itself).
        float4 wa = w4[0], wb = w4[1];
        const cf* q = reinterpret_cast<const cf*>(p);
        cf p0 = q[0], p1 = q[1], p2 = q[2], p3 = q[3];
        for (int j = 0; j < d.z; ++j) {
            const int jn = j + 1 < d.z ? j + 1 : j;
            const float4 nwa = w4[2 * jn], nwb = w4[2 * jn + 1];
            const cf* qn = reinterpret_cast<const cf*>(p + 8 * jn);
            const cf n0 = qn[0], n1 = qn[1], n2 = qn[2], n3 = qn[3];
            acc0 = __builtin_elementwise_fma(mkc(wa.x, wa.y), p0, acc0);
            acc1 = __builtin_elementwise_fma(mkc(wa.z, wa.w), p1, acc1);
            acc0 = __builtin_elementwise_fma(mkc(wb.x, wb.y), p2, acc0);
            acc1 = __builtin_elementwise_fma(mkc(wb.z, wb.w), p3, acc1);
            wa = nwa; wb = nwb; p0 = n0; p1 = n1; p2 = n2; p3 = n3;
        }
        orow[d.x] = (acc0.x + acc0.y) + (acc1.x + acc1.y);
    }
}


}  // namespace tac
