// Balanced launch schedules for mid-size batches (host side, no device code).
//
// One launch of the factorisation (block column k) is a bag of tiles of very different length: panel tiles and the
// diagonal look-ahead carry k K-blocks, the trtri tiles of row k-1 carry 1 .. k-1, the diagonal blocks a fixed latency
// chain.  With 64 matrices a launch has > 2000 tiles and the hardware dispatcher balances them; with 8 it has 264 for
// 256 CUs, all resident at once, and lasts as long as its longest tile while the CUs holding short tiles idle (B = 8,
// N = 4096: 4.6 ms where the same work spread evenly would take 2.6).  Here the host deals the work out itself: long
// tiles are cut into K-slices no longer than a fraction of the mean load per workgroup (the split-K slab machinery
// sums them), every piece gets a cost in K-block units, and the pieces are put in the grid LONGEST FIRST, one workgroup
// each (factor_step_sched_kernel).  The kernel is launched with enough LDS that one workgroup fits a CU, so the
// hardware dispatcher hands the next piece to the first CU that falls free: list scheduling in LPT order.  (A first
// version ran G persistent workgroups over host-assigned lists; with every piece type inlined in one loop the compiler
// hoisted the address arithmetic of all of them out of it -- 540 spilled VGPRs.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <queue>
#include <vector>

namespace volt {

enum SchedKind { SK_DIAG = 0, SK_LOOKAHEAD = 1, SK_PANEL = 2, SK_TRTRI = 3, SK_TRTRI_DIAG = 4 };

struct SchedItem {           // 16 bytes, read as one int4 on the device
    int kind_b;              // kind | b << 3
    int idx;                 // panel: block row i;  trtri: block column j
    int slice;               // sl | nsl << 8 | tile << 16   (tile: slab / counter slot of the launch, < B (n+1))
    int range;               // b0 | b1 << 16  (K blocks of this slice)
};

struct SchedParams {
    int G = 256;             // workgroups resident at a time (CUs x occupancy): sets the mean load the slices are sized by
    int S = 4;               // most slices a tile may be cut into (slab slots per tile)
    float frac = 0.6f;       // a slice is at most frac * (mean load per workgroup) long ...
    int min_len = 2;         // ... but never shorter than this many K blocks
    float diag_cost = 3.2f;  // diagonal block (update of the last K block + factorisation + inverse), in K-block units
    float phase2 = 1.0f;     // the W product + store of a panel / trtri tile
    float dump = 0.25f;      // writing one slab / adding one slab up
};

// Launch k of a factorisation of B matrices of n block columns (k == n: the trailing launch that carries only the
// last trtri row).  Appends the items in grid order: the diagonal blocks (the panel tiles of the launch wait for them),
// then everything else by falling cost.  loads (optional): the load of each of G slots under greedy list scheduling
// of that order, in K-block units -- what the dispatcher is expected to do.
inline void sched_build_launch(int B, int n, bool has_y, int k, const SchedParams& p, std::vector<SchedItem>& items,
                               std::vector<float>* loads = nullptr) {
    struct Piece { SchedItem it; float cost; };
    struct Tile { int kind, b, idx, kb, id; float tail; };
    std::vector<Tile> tiles;
    std::vector<Piece> diag, rest;
    const int itri = has_y ? (k < n ? k - 1 : n - 1) : -1;
    if (k < n) {
        for (int b = 0; b < B; ++b)
            diag.push_back({{SK_DIAG | b << 3, k, 0, 0}, k > 0 ? p.diag_cost : p.diag_cost - 1.f});
        if (k >= 1 && k + 1 < n)
            for (int b = 0; b < B; ++b) tiles.push_back({SK_LOOKAHEAD, b, k + 1, k, b, 0.2f});
        for (int t = 0; t < n - k - 1; ++t)
            for (int b = 0; b < B; ++b) tiles.push_back({SK_PANEL, b, k + 1 + t, k, B + t * B + b, p.phase2});
    }
    const int npan = k < n ? (n - k - 1) * B : 0;
    if (itri >= 0)
        for (int j = 0; j <= itri; ++j)
            for (int b = 0; b < B; ++b) {
                if (j == itri) rest.push_back({{SK_TRTRI_DIAG | b << 3, j, 0, 0}, 0.7f});
                else tiles.push_back({SK_TRTRI, b, j, itri - j, B + npan + j * B + b, p.phase2});
            }
    double total = 0;
    for (const Piece& d : diag) total += d.cost;
    for (const Piece& d : rest) total += d.cost;
    for (const Tile& t : tiles) total += t.kb + t.tail;
    const float mean = (float)(total / p.G);
    const int lmax = std::max(p.min_len, (int)std::ceil(p.frac * mean));
    // `tiles` holds B consecutive entries per tile position (same kind, index and length, b = 0 .. B-1).  The slices of
    // a position go out slice by slice with b innermost, so that B consecutive workgroups are the B matrices' copies of one
    // piece: the dispatcher puts workgroup w on XCD w % 8, and with B a multiple of 8 every XCD then keeps working on
    // the same matrices (their shared block row stays in ITS L2) -- as decode_tile_batch arranges for the plain launches.
    for (size_t c0 = 0; c0 < tiles.size(); c0 += (size_t)B) {
        const Tile& t0 = tiles[c0];
        int nsl = (t0.kb + lmax - 1) / lmax;
        nsl = std::max(1, std::min(nsl, p.S));
        for (int sl = 0; sl < nsl; ++sl) {
            const int b0 = sl * t0.kb / nsl, b1 = (sl + 1) * t0.kb / nsl;
            float cost = (float)(b1 - b0) + t0.tail / nsl;
            if (nsl > 1) cost += p.dump + p.dump;                // its own dump + its share of the summing
            for (int b = 0; b < B; ++b) {
                const Tile& t = tiles[c0 + b];
                rest.push_back({{t.kind | t.b << 3, t.idx, sl | nsl << 8 | t.id << 16, b0 | b1 << 16}, cost});
            }
        }
    }
    std::stable_sort(rest.begin(), rest.end(), [](const Piece& a, const Piece& b) { return a.cost > b.cost; });
    for (const Piece& d : diag) items.push_back(d.it);
    for (const Piece& pc : rest) items.push_back(pc.it);
    if (loads) {
        typedef std::pair<float, int> Slot;                      // (load, slot), least loaded on top
        std::priority_queue<Slot, std::vector<Slot>, std::greater<Slot>> heap;
        for (int g = 0; g < p.G; ++g) heap.push({0.f, g});
        loads->assign(p.G, 0.f);
        auto place = [&](float cost) {
            Slot sl = heap.top();
            heap.pop();
            sl.first += cost;
            (*loads)[sl.second] = sl.first;
            heap.push(sl);
        };
        for (const Piece& d : diag) place(d.cost);
        for (const Piece& pc : rest) place(pc.cost);
    }
}

}  // namespace volt
