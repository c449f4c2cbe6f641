// The piece list of the one-launch step for ONE long series (long_step_kernel in chol.hip): host side, no device code.
//
// The one-launch step of short series (small_step_kernel) makes every tile a workgroup that starts when the flags of what
// it reads say so.  A tile deep in a long matrix has a long first phase and learns of its last inputs only a block column
// before it is due, so here the EARLY part of a first phase (all K blocks but the last, whose operand the previous spine
// hands on) is cut into slices, each a piece of its own placed right behind the block column that completes its K range:
// it runs ahead of its tile, dumps its partial accumulators into a slab (write-through), and the tile's own piece only
// adds the slabs up.  Slices grow GEOMETRICALLY away from the tile -- the last one `first` blocks, then 2 first, 4 first,
// .. -- so that the slice that becomes ready one block column before its tile is short, the long ones have block columns
// of slack, and every block column completes about the same number of slices (~75 at 32 block columns): the launch runs
// one workgroup per CU (a pivot chain that shares its CU runs 1.5 - 3x slower), i.e. 256 resident workgroups, and what
// must be resident at a time is two groups of tiles and two groups of slices.
// The list is in dependency order (a piece depends only on pieces before it; workgroups are dispatched in grid order).
// Group g = block column g:
//     D(0) | S(g)                 the first diagonal block | the spine: tile (g,g-1), then diagonal block g
//     P(i,g), i = g+2 .. n-1      the other panel tiles of the column
//     T(g-1, j), j = 0 .. g-1     row g-1 of the inverse (the diagonal tile last)
//     U(g+1)                      the look-ahead part of A[g+1,g+1]: its early blocks (m < g-1), all slices, summed and parked;
//                                 the last block (m = g-1) is a slab from P(g+1,g-1) that the spine S(g+1) adds itself
//     E(..)                       the slices whose K range this block column completes
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace volt {

enum LongKind { LG_D0 = 0, LG_SPINE = 1, LG_PANEL = 2, LG_U = 3, LG_T = 4, LG_TDIAG = 5, LG_E_PANEL = 6, LG_E_T = 7, LG_E_U = 8,
                LG_R = 9 };   // R(g): the second half of a split spine (rank-32 updates of the diagonal tile + diagonal block g)

struct LongItem {            // 16 bytes, read as one int4 on the device
    int kind_ij;             // kind | a << 8 | b << 16:  D/S/U: a = k;  P, E_PANEL: a = tile row, b = diagonal block above;  T, E_T: a = i, b = j
    int slice;               // E pieces: b0 | b1 << 8 (K blocks [b0, b1) of the tile's first phase);  base pieces: number of slices (0: none)
    int slab;                // E pieces: their slab slot;  sliced base pieces: their first slab slot (slices are consecutive)
    int cnt;                 // the tile's slice counter (index), -1 if it has no slices
};

struct LongPlan {
    std::vector<LongItem> items;
    std::vector<LongItem> uinfo;   // [n] per diagonal tile k: {U(k) exists, -, slab, counter} of the last block of its look-ahead part,
                                   // which P(k,k-2) contributes (its own tile times its transpose) the moment the tile is there
    int nslabs = 0;          // slab slots (128 x 128 floats each)
    int ncnt = 0;            // slice counters
    int xcd_from = 0;        // > 0: D(0) and every spine S(g), g >= xcd_from, sit at a grid index that is a multiple of 8
};

// xcd_align: workgroup i runs on XCD i % 8, and a hand-off between workgroups of ONE XCD can be served by that XCD's L2
// instead of a trip through the fabric.  The chain's hand-offs are spine to spine, so the spines are put at grid indices
// that are multiples of 8: up to 7 tiles of row g-1 of the inverse (which do not depend on S(g)) go in front of S(g).
// split_spine: S(g) only solves its tile (g,g-1) by substitution and hands its 32-column slabs on; R(g), right behind it in
// the list and on another CU, takes the rank-32 updates of the diagonal tile and diagonal block g.
inline LongPlan long_build(int n, int first, int emin, bool xcd_align = false, bool split_spine = false) {
    LongPlan pl;
    struct Slice { LongItem it; int ready; };
    std::vector<Slice> slices;
    auto pack = [](int kind, int a, int b) { return kind | a << 8 | b << 16; };
    // a tile's early part of `eb` blocks: returns the base piece's (nslices, first slab, counter) and queues the E pieces
    pl.uinfo.assign(n, LongItem{0, 0, 0, -1});
    auto cut = [&](int ekind, int a, int b, int eb, auto ready_of, int emin_here = -1) {
        LongItem base{0, 0, 0, -1};
        if (eb <= (emin_here >= 0 ? emin_here : emin)) return base;
        std::vector<int> ends;                                   // slice ends, from the tile backwards: eb, eb - first, eb - 3 first, ..
        for (int end = eb, len = first; end > 0; end -= len, len *= 2) ends.push_back(end);
        std::reverse(ends.begin(), ends.end());
        base.slice = (int)ends.size();
        base.slab = pl.nslabs;
        base.cnt = pl.ncnt++;
        int b0 = 0;
        for (int b1 : ends) {
            slices.push_back({{pack(ekind, a, b), b0 | b1 << 8, pl.nslabs++, base.cnt}, ready_of(b1)});
            b0 = b1;
        }
        return base;
    };
    std::vector<std::vector<LongItem>> groups(n + 1);
    for (int g = 0; g <= n; ++g) {
        std::vector<LongItem>& G = groups[g];
        if (g == 0) G.push_back({pack(LG_D0, 0, 0), 0, 0, -1});
        else if (g < n) {
            // spine tile (g, g-1) under diagonal block kd = g-1: K blocks 0 .. kd-1, early part kd-1 blocks; an early slice
            // [b0,b1) reads L[g, <b1] and L[g-1, <b1], panel tiles of column b1-1: there behind group b1-1
            LongItem it = cut(LG_E_PANEL, g, g - 1, g - 2, [](int b1) { return b1 - 1; });
            it.kind_ij = pack(LG_SPINE, g, 0);
            G.push_back(it);
            if (split_spine) G.push_back({pack(LG_R, g, 0), 0, 0, -1});
        }
        for (int i = g + 2; i < n; ++i) {   // P(i,g): K blocks 0 .. g-1, early part g-1 blocks
            LongItem it = cut(LG_E_PANEL, i, g, g - 1, [](int b1) { return b1 - 1; });
            it.kind_ij = pack(LG_PANEL, i, g);
            G.push_back(it);
        }
        if (g >= 1) {
            const int i = g - 1;
            for (int j = 0; j < i; ++j) {   // T(i,j): K blocks m = j .. i-1, early part i-j-1 blocks; slice [b0,b1) reads
                                            // L[i, j+b0 .. j+b1-1] and Y[j, same columns] = T(j+b1-1, j): group j+b1
                LongItem it = cut(LG_E_T, i, j, i - j - 1, [j](int b1) { return j + b1; });
                it.kind_ij = pack(LG_T, i, j);
                G.push_back(it);
            }
            G.push_back({pack(LG_TDIAG, i, i), 0, 0, -1});
        }
        if (g + 1 >= 2 && g + 1 <= n - 1) { // the look-ahead part of A[k,k], k = g+1 (blocks 0 .. k-2 of its update):
            const int k = g + 1;            //   U(k): the k-2 early blocks, ALL as slices, summed and parked in A  (k >= 3)
            LongItem u{0, 0, 0, -1};        //   the last block (L[k,k-2], out with this group): a slab from P(k,k-2) itself, which
            if (k >= 3) {                   //   the spine S(k) adds when it loads its accumulators
                u = cut(LG_E_U, k, 0, k - 2, [](int b1) { return b1 - 1; }, 0);
                u.kind_ij = pack(LG_U, k, 0);
                G.push_back(u);
            }
            pl.uinfo[k] = {k >= 3 ? 1 : 0, 0, pl.nslabs++, pl.ncnt++};      // {U(k) exists, -, P's slab, its counter}
        }
    }
    int unaligned_last = 0;                                      // the last spine that could not be aligned
    for (int g = 0; g <= n; ++g) {
        std::vector<LongItem>& G = groups[g];
        size_t lead = 0;
        const bool spine = g >= 1 && g < n;
        if (xcd_align && spine) {
            const size_t need = (8 - pl.items.size() % 8) % 8;
            // fillers: T(g-1, j) pieces (kind LG_T), from the front of the row (the longest tiles)
            std::vector<size_t> fill;
            for (size_t q = 1; q < G.size() && fill.size() < need; ++q)
                if ((G[q].kind_ij & 0xff) == LG_T) fill.push_back(q);
            if (fill.size() == need) {
                for (size_t q : fill) pl.items.push_back(G[q]);
                for (size_t q = fill.size(); q-- > 0;) G.erase(G.begin() + (long)fill[q]);
            } else {
                unaligned_last = g;
            }
        }
        (void)lead;
        for (const LongItem& it : G) pl.items.push_back(it);
        for (const Slice& s : slices)
            if ((s.ready < 0 ? 0 : s.ready) == g) pl.items.push_back(s.it);
    }
    pl.xcd_from = xcd_align ? unaligned_last + 1 : 0;
    if (pl.xcd_from >= n) pl.xcd_from = 0;
    return pl;
}

}  // namespace volt
