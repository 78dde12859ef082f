// ola_runs.hpp — which hop-runs of a row the overlap-adding backward kernels leave to the fold kernel (plain C++: the host
// launch code of backward.hip uses it, tests/test_host_api.py compiles it for the host and checks it against ola_direct()'s rule).
#pragma once

namespace tac {

// Run fc of a row = padded positions [fc hop, (fc + 1) hop).  Left to ola_fold_runs_kernel: runs 0 .. head - 1 (padding images at
// the row's start), runs tail_first .. last_run (padding images at its end, runs past the last frame) and, of every segment but
// the first, its first zone_frames runs (they still lack the previous segment's edge sums) — unless already in head / tail.
struct OlaRuns {
    int head;          // runs 0 .. head - 1
    int tail_first;    // runs tail_first .. last_run
    int last_run;      // the run of the row's last sample
    int zone_frames;   // ceil((N - hop) / hop)
};

// plain_pad: no centre padding or constant padding (no images: a run is clean when it lies inside the row); otherwise a clean run
// keeps clear of the `pad` samples at either end that own reflect / replicate / circular images.
inline OlaRuns ola_runs_for(int length, int pad, int hop, int n_fft, int n_frames, bool plain_pad) {
    OlaRuns r;
    r.last_run = (length + pad - 1) / hop;
    int first_clean = plain_pad ? (pad + hop - 1) / hop : (2 * pad) / hop + 1;
    const int last_clean = plain_pad ? (length + pad - hop >= 0 ? (length + pad - hop) / hop : -1)
                                     : (length - 1 - hop >= 0 ? (length - 1 - hop) / hop : -1);
    if (first_clean > r.last_run + 1) first_clean = r.last_run + 1;
    r.head = first_clean;
    const int last_direct = last_clean < n_frames - 1 ? last_clean : n_frames - 1;     // frames past T - 1 are never stored directly
    r.tail_first = first_clean > last_direct + 1 ? first_clean : last_direct + 1;
    r.zone_frames = (n_fft - hop + hop - 1) / hop;
    return r;
}

}  // namespace tac
