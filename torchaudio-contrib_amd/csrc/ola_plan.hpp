// ola_plan.hpp — segmentation shared by the backward kernels that overlap-add inside the kernel (backward.hip,
// backward_ring3*.hpp, stft_n400.hip) and ola_fold_kernel, which finishes segment borders and padding images.
#pragma once
#include "host_common.hpp"

namespace tac {

struct OlaPlan {
    int seg_frames;       // S: frames per segment (>= (N - hop) / hop, so that a tail never reaches past the next segment)
    int segs_per_row;
    long long pad_len;    // floats per row of gpad (= length + 2·center_pad)
    int n_fft;
    int direct;           // the fft_length-2048 kernel stores the clean interior straight into the waveform gradient
    float* gwave;         // ... here (row r at gwave + r * gstride)
    long long gstride;
};

// Frame f's `hop` complete positions need nothing but themselves: they lie outside the border zone of their segment
// (whose first N - hop positions still lack the previous segment's edge sums) and no sample among them has a reflect /
// replicate / circular image or falls into the padding.  Such runs go straight into the waveform gradient; the fold
// kernel only handles the rest (round 3: it used to copy the whole padded gradient, 0.064 ms at cfg-2).
__device__ __forceinline__ bool ola_direct(const FrameGeom& g, const OlaPlan& plan, int f) {
    if (!plan.direct) return false;
    const int S = plan.seg_frames, hop = g.hop, pad = g.center_pad, L = (int)g.length;
    const int sg = f / S;
    if (sg > 0 && (f - sg * S) * hop < plan.n_fft - hop) return false;
    const int jlo = f * hop - pad, jhi = jlo + hop - 1;
    if (pad == 0 || g.pad_mode == PAD_CONSTANT) return jlo >= 0 && jhi < L;
    return jlo > pad && jhi < L - 1 - pad;
}

}  // namespace tac
