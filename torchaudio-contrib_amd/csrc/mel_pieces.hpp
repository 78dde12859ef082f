// mel_pieces.hpp — "piece" layout of a band-sparse filterbank for the fft_length-2048 streaming kernel (host side; plain C++,
// no HIP: tests/test_host_api.py drives it through tac_melbank_plan_pieces_host).
//
// The classic lane layout (pack_lanes, melspec_sparse.hip) gives lane l the whole bands l and 64 + l: every lane runs as many
// four-tap steps as the WIDEST band of each slot — 4 + 14 = 18 for the standard 128-band bank, whose bands need 601 quads in
// all, 9.4 per lane.  Here a lane runs THREE segments of L0 / L1 / L2 steps (12 for that bank), each holding a PIECE of a band:
// a band of nq quads takes ceil(nq / L_s) <= 3 consecutive pieces in ADJACENT lanes of one 16-lane row of one segment.  After the
// contraction the pieces are summed with row-shift DPP reads (a lane adds its left neighbour's / second-left neighbour's partial
// sum when its piece index is >= 1 / >= 2), the lane of a band's LAST piece writes the total into the band's slot of a staging
// row in LDS, and lane l reads back bands l and 64 + l for the epilogue and two coalesced row stores.  (Storing from the last
// piece's lane directly — three scattered, masked global stores — cost 4.3 % of the kernel; adding the partial sums into the
// staging row with ds_add_f32 instead of the DPP reads cost 54 %: tools/ablation/README.md.)
// STATUS (round 4): correct (the GPU suite passes through it) and NOT faster — 0.1179 vs 0.1125 ms (+4.6 %) on the standard bank, although
// the same kernel with 12 contraction steps and no piece bookkeeping measures -5.3 %: the cross-lane sums, the staging round trip and
// the 168 registers they need cost more than the six saved steps.  Opt-in (TAC_MEL_PIECES=1); the classic layout ships.
// Reference: functional.py:172-184 (apply_filterbank) for the standard banks of functional.py:131-169.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

namespace tac {

constexpr int MP_SEGS = 3;
constexpr int MP_MAX_PIECES = 3;                 // pieces per band (two shifted adds)

struct PiecePlan {
    int L[MP_SEGS] = {0, 0, 0};                  // four-tap steps of the three segments
    int total_steps = 0;
    std::vector<int32_t> first;                  // [3][64] first bin (multiple of 4) the lane's segment reads
    std::vector<int32_t> band;                   // [3][64] band the lane-segment's piece belongs to, or -1 (unused)
    std::vector<int32_t> index;                  // [3][64] position of the piece inside its band (0 / 1 / 2), + 256 on the band's LAST piece
    std::vector<float> w;                        // [total_steps][64][4] zero-padded weights
};

// candidate segment lengths, shortest total first
static const int MP_CANDIDATES[][MP_SEGS] = {{3, 4, 5}, {4, 5, 6}};          // (== the kernel instantiations of launch_stream)
constexpr int MP_NCAND = 2;

// the sixteen lanes the LDS serves together for a 16-byte read (MI355X_MICROARCH.md, ds_read_b128): group gi of a wave
inline void mp_b128_group(int gi, int (&lanes)[16]) {
    static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    for (int i = 0; i < 16; ++i) lanes[i] = kGroup[gi & 1][i] + 32 * (gi >> 1);
}

// h: dense (n_freqs, n_mels) row-major bank; limit: floats of a row buffer (bins + zeroed slack, multiple of 4).
// Returns false when no candidate fits (the caller keeps the classic layout).
inline bool plan_pieces(const float* h, int n_freqs, int n_mels, int limit, PiecePlan* out) {
    if (n_mels < 1 || n_mels > MP_SEGS * 64) return false;
    std::vector<int> q0(n_mels, 0), nq(n_mels, 0), blo(n_mels, 0), bhi(n_mels, 0);
    for (int m = 0; m < n_mels; ++m) {
        int l0 = n_freqs, h0 = 0;
        for (int f = 0; f < n_freqs; ++f)
            if (h[(size_t)f * n_mels + m] != 0.0f) { l0 = f < l0 ? f : l0; h0 = f + 1; }
        if (h0 > l0) {
            blo[m] = l0;
            bhi[m] = h0;
            q0[m] = l0 / 4;
            nq[m] = (h0 + 3) / 4 - q0[m];
        }
    }
    std::vector<int> order(n_mels);
    for (int m = 0; m < n_mels; ++m) order[m] = m;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nq[a] > nq[b]; });
    for (int ci = 0; ci < MP_NCAND; ++ci) {
        const int* L = MP_CANDIDATES[ci];
        int fill[MP_SEGS][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        struct Place { int seg, lane0, np; };
        std::vector<Place> place(n_mels);
        bool ok = true;
        for (int oi = 0; oi < n_mels && ok; ++oi) {
            const int m = order[oi];
            const int need = nq[m] > 0 ? nq[m] : 1;                 // (an empty band still needs a lane to store its zero)
            // fewest pieces first, then the shortest segment that achieves it; best-fit row (the fullest that still takes it)
            int opts[MP_SEGS][3], nopt = 0;
            for (int s = 0; s < MP_SEGS; ++s) {
                const int np = (need + L[s] - 1) / L[s];
                if (np <= MP_MAX_PIECES) { opts[nopt][0] = np; opts[nopt][1] = L[s]; opts[nopt][2] = s; ++nopt; }
            }
            for (int a = 0; a < nopt; ++a)
                for (int b = a + 1; b < nopt; ++b)
                    if (opts[b][0] < opts[a][0] || (opts[b][0] == opts[a][0] && opts[b][1] < opts[a][1]))
                        for (int c = 0; c < 3; ++c) std::swap(opts[a][c], opts[b][c]);
            bool placed = false;
            for (int a = 0; a < nopt && !placed; ++a) {
                const int np = opts[a][0], s = opts[a][2];
                int best = -1;
                for (int r = 0; r < 4; ++r)
                    if (fill[s][r] + np <= 16 && (best < 0 || fill[s][r] > fill[s][best])) best = r;
                if (best >= 0) {
                    place[m] = Place{s, 16 * best + fill[s][best], np};
                    fill[s][best] += np;
                    placed = true;
                }
            }
            ok = placed;
        }
        if (!ok) continue;
        // ---- lay the plan out
        PiecePlan& p = *out;
        p.total_steps = 0;
        int base[MP_SEGS];
        for (int s = 0; s < MP_SEGS; ++s) {
            p.L[s] = L[s];
            base[s] = p.total_steps;
            p.total_steps += L[s];
        }
        p.first.assign(MP_SEGS * 64, 0);
        p.band.assign(MP_SEGS * 64, -1);
        p.index.assign(MP_SEGS * 64, 0);
        p.w.assign((size_t)p.total_steps * 256, 0.0f);
        std::vector<int> dq(MP_SEGS * 64, 0), dlen(MP_SEGS * 64, 0), owner(MP_SEGS * 64, -1);   // data quads of a lane-segment
        for (int m = 0; m < n_mels; ++m) {
            const Place& pl = place[m];
            for (int i = 0; i < pl.np; ++i) {
                const int e = pl.seg * 64 + pl.lane0 + i;
                owner[e] = m;
                p.index[e] = i + (i == pl.np - 1 ? 256 : 0);
                dq[e] = q0[m] + i * L[pl.seg];
                const int left = nq[m] - i * L[pl.seg];
                dlen[e] = left > L[pl.seg] ? L[pl.seg] : (left > 0 ? left : 0);
            }
        }
        for (int e = 0; e < MP_SEGS * 64; ++e) p.band[e] = owner[e];
        // ---- starts: a lane-segment reads L quads from `first`; its data may sit anywhere inside (zeros around it), the whole
        //      run must stay inside the row, and within every sixteen-lane b128 group the starts should fall into different
        //      16-byte bank groups ((bin / 4) mod 16): bipartite matching lane -> residue with the smallest load per residue
        const int limq = limit / 4;
        for (int s = 0; s < MP_SEGS; ++s)
            for (int gi = 0; gi < 4; ++gi) {
                int lanes[16];
                mp_b128_group(gi, lanes);
                std::vector<int> cand[16];
                for (int i = 0; i < 16; ++i) {
                    const int e = s * 64 + lanes[i];
                    int lo_q, hi_q;                                     // admissible first quads [lo_q, hi_q]
                    if (owner[e] < 0 || dlen[e] == 0) { lo_q = 0; hi_q = 15 < limq - L[s] ? 15 : limq - L[s]; }
                    else {
                        hi_q = dq[e];
                        lo_q = dq[e] + dlen[e] - L[s];
                    }
                    if (hi_q > limq - L[s]) hi_q = limq - L[s];
                    if (lo_q < 0) lo_q = 0;
                    if (lo_q > hi_q) lo_q = hi_q;                       // (cannot happen: data ends inside the row)
                    for (int c = hi_q; c >= lo_q; --c) cand[i].push_back(c);
                }
                int choice[16];
                for (int cap = 1; cap <= 16; ++cap) {
                    std::vector<int> load[16];
                    unsigned seen = 0;
                    std::function<bool(int)> put = [&](int i) -> bool {
                        for (int c : cand[i]) {
                            const int r = c & 15;
                            if (seen & (1u << r)) continue;
                            seen |= 1u << r;
                            if ((int)load[r].size() < cap) { load[r].push_back(i); choice[i] = c; return true; }
                            for (size_t k = 0; k < load[r].size(); ++k) {
                                const int other = load[r][k];
                                if (put(other)) {
                                    load[r].erase(std::find(load[r].begin(), load[r].end(), other));
                                    load[r].push_back(i);
                                    choice[i] = c;
                                    return true;
                                }
                            }
                        }
                        return false;
                    };
                    bool all = true;
                    for (int i = 0; i < 16 && all; ++i) {
                        seen = 0;
                        all = put(i);
                    }
                    if (all) break;
                }
                for (int i = 0; i < 16; ++i) p.first[s * 64 + lanes[i]] = 4 * choice[i];
            }
        // ---- weights
        for (int s = 0; s < MP_SEGS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int e = s * 64 + l, m = owner[e];
                if (m < 0) continue;
                for (int j = 0; j < L[s]; ++j)
                    for (int u = 0; u < 4; ++u) {
                        const int bin = p.first[e] + 4 * j + u;
                        const bool mine = bin >= 4 * dq[e] && bin < 4 * (dq[e] + dlen[e]) && bin >= blo[m] && bin < bhi[m] && bin < n_freqs;
                        p.w[((size_t)(base[s] + j) * 64 + l) * 4 + u] = mine ? h[(size_t)bin * n_mels + m] : 0.0f;
                    }
            }
        return true;
    }
    return false;
}

}  // namespace tac
