// The piece list of the one-launch batched step (host side, no device code; the kernel is batch_step.hip).
//
// One MLL + gradient step of B matrices of n block columns is B * (n (n + 3) / 2 + ...) tiles with a fixed dependency
// graph.  The launch-per-column schedules (chol.hip) cut that graph at every block column: 32 launches, each as long as
// its longest tile plus a tail.  Here the whole step is ONE launch whose workgroups pull the pieces of this list by ticket
// (common.h, "who runs which piece").  The list is in TOPOLOGICAL order: a piece only waits for pieces listed before it, a
// ticket is taken by a running workgroup, so whatever is waited for is running or finished -- in whatever order the
// dispatcher starts workgroups; tiles do not wait for their inputs before they start but chase them K block by K block
// (common.h, Chase).
//
// Order: for k = 0 .. n-1 the pieces of block column k --
//     D(k)        diagonal tile (k,k): last K block, factor, invert -> W_k
//     LA(k)       1 <= k <= n-2: look-ahead, A[k+1,k+1] -= sum_{m<k} L[k+1,m] L[k+1,m]^T
//     P(i,k)      i = k+1 .. n-1: panel tile, update + solve
//     TD(k-1), T(k-1,j), j = 0 .. k-2     row k-1 of the triangular inverse (if wanted), longest tile first
//     AL(k-2)     block k-2 of z = Y'r (the sum of row k-2's z-partials) and alpha's partial sums from block column k-2
//                 of Y: an HBM / L2 stream that rides beside the MFMA tiles (mll.hip's sum_zpart / y_times_z launches)
// and finally row n-1 of the inverse with AL(n-2), AL(n-1).  Every position is emitted for all B matrices in groups of 8 that
// alternate: with B a multiple of 8 piece w belongs to a matrix = w (mod 8), and ticket t of queue q is piece 8 t + q -- queue q is
// the matrices = q (mod 8), claimed and run by ONE XCD (the pullers read their XCC id), so a matrix lives under one L2: its
// shared block rows in it, its hand-offs through it.
#pragma once
#include <cstdint>
#include <vector>

namespace volt {

enum BatchKind { BK_DIAG = 0, BK_LOOKAHEAD = 1, BK_PANEL = 2, BK_TRTRI = 3, BK_TRTRI_DIAG = 4, BK_ALPHA = 5 };

struct BatchItem {           // 16 bytes, read as one int4 on the device
    int kind_b;              // kind | b << 3
    int row;                 // D: k;  LA: k (the tile is (k+1,k+1));  P: i;  T / TD: i;  AL: c
    int col;                 // P: k;  T: j;  AL: which chunk of BATCH_ALPHA_ROWS rows;  else 0
    int pad;                 // the table's check word (batch_step.hip), the same in every piece
};

constexpr int BATCH_MAGIC = 0x564f4c42;        // "VOLB"
constexpr int BATCH_ALPHA_ROWS = 1024;         // rows of Y per alpha piece: block column c of Y has (c + 1) * 128 of them
constexpr int BATCH_HDR = 16;                  // int4 slots ahead of the items (two are used: the table's identity)

// pieces of one step (closed form: the step itself needs no host table to size its grid)
inline int64_t batch_count(int B, int n, bool has_y) {
    int64_t per = 0;
    for (int k = 0; k < n; ++k) per += 1 + ((k >= 1 && k + 1 < n) ? 1 : 0) + (n - k - 1) + ((has_y && k >= 1) ? k : 0);
    if (has_y) {
        per += n;                              // the last row of the inverse
        for (int c = 0; c < n; ++c) per += ((c + 1) * 128 + BATCH_ALPHA_ROWS - 1) / BATCH_ALPHA_ROWS;   // the alpha pieces
    }
    return per * B;
}

// order: 0 = positions in the order above, matrix innermost;  1 = panel / trtri tiles of a block column matrix-major in
// groups of 8 matrices (the 8 matrices of a group still alternate, so the XCD mapping holds): a matrix's tiles of one
// block column are then adjacent in ITS XCD's queue and share its block row k while it is hot
// lad: how many block columns EARLY the look-ahead tiles are listed.  LA(k) -- A[k+1,k+1] -= sum_{m<k} L[k+1,m] L[k+1,m]^T, k K
// blocks on the 2x2-wave core -- feeds D(k+1); listed with column k it starts when D(k) is dispatched and, sharing its CU,
// needs ~14 us per block: with few matrices per XCD it finishes AFTER W_k and the pivots wait for it (8 x 4096: LA(20) ends
// 117 us after D(20), stamps of profiles/r05).  Listed lad columns earlier it chases row k+1 from then on and has one block
// left when column k-1 completes.  It then waits for a piece listed AFTER it (P(k+1,k-1)): allowed because only lad * B
// workgroups at a time do so -- the caller keeps that far below the number of resident workgroups -- so the pieces they wait
// for are always dispatched (the dispatch-order argument needs every waiter to wait on a RUNNING piece; here the few
// forward waiters cannot fill the chip).
inline void batch_build(int B, int n, bool has_y, int order, std::vector<BatchItem>& items, int lad = 0) {
    auto emit = [&](int kind, int row, int col) {
        for (int b = 0; b < B; ++b) items.push_back({kind | b << 3, row, col, 0});
    };
    // `pos` positions (kind, row(p), col(p)) for all matrices: position-major (order 0) or, per group of 8 matrices,
    // matrix-group-major (order 1)
    auto emit_run = [&](int kind, int npos, auto rowf, auto colf) {
        if (order == 0 || (B & 7)) {
            for (int p = 0; p < npos; ++p) emit(kind, rowf(p), colf(p));
            return;
        }
        if (order == 1) {
            for (int g = 0; g < B / 8; ++g)
                for (int p = 0; p < npos; ++p)
                    for (int x = 0; x < 8; ++x) items.push_back({kind | (g * 8 + x) << 3, rowf(p), colf(p), 0});
            return;
        }
        // order = 2 + 4 * (log2 of the window) + 32 * (groups of 8 matrices side by side - 1)  (round 6): positions in
        // WINDOWS of `win`, and inside a window `side` groups of 8 matrices at a time -- a queue (one XCD) then sees `side` of
        // its matrices x `win` consecutive positions back to back, so that the tiles resident on an XCD share the block row k of
        // FEWER matrices (its 64 pullers: side x win tiles; order 0: 8 matrices x 8 positions)
        const int win = 1 << ((order >> 2) & 7), side = ((order >> 5) & 7) + 1, G = B / 8;
        for (int p0 = 0; p0 < npos; p0 += win)
            for (int g0 = 0; g0 < G; g0 += side)
                for (int p = p0; p < npos && p < p0 + win; ++p)
                    for (int g = g0; g < G && g < g0 + side; ++g)
                        for (int x = 0; x < 8; ++x) items.push_back({kind | (g * 8 + x) << 3, rowf(p), colf(p), 0});
    };
    auto alpha = [&](int c) {
        for (int q = 0; q * BATCH_ALPHA_ROWS < (c + 1) * 128; ++q) emit(BK_ALPHA, c, q);
    };
    auto trtri_row = [&](int i) {
        emit(BK_TRTRI_DIAG, i, 0);
        emit_run(BK_TRTRI, i, [&](int p) { return i; }, [&](int p) { return p; });
    };
    for (int k = 0; k < n; ++k) {
        emit(BK_DIAG, k, 0);
        for (int kk = 1; kk + 1 < n; ++kk)                   // the look-ahead tiles listed with this column
            if ((kk - lad > 1 ? kk - lad : 1) == k) emit(BK_LOOKAHEAD, kk, 0);
        // the row the next diagonal tile needs first, for every matrix; then the rest
        if (k + 1 < n) emit(BK_PANEL, k + 1, k);
        emit_run(BK_PANEL, n - k - 2 > 0 ? n - k - 2 : 0, [&](int p) { return k + 2 + p; }, [&](int p) { return k; });
        if (has_y && k >= 1) trtri_row(k - 1);
        if (has_y && k >= 2) alpha(k - 2);
    }
    if (has_y) {
        trtri_row(n - 1);
        if (n >= 2) alpha(n - 2);
        alpha(n - 1);
    }
}

// progress words per matrix (ints): rowp[n] | tcol[n] | la[n], padded to a multiple of 32
inline int batch_pstride(int n) { return (3 * n + 31) & ~31; }

}  // namespace volt
