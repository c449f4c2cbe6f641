"""The piece list of the one-launch batched step (csrc/batch_sched.h, through the host-only hook volt_batch_describe):
every piece sits behind everything it waits for -- the property the in-launch hand-offs rest on (workgroups are
dispatched in grid order, so what a resident workgroup waits for is resident or finished) --, every tile of the
factorisation / inverse appears exactly once, and with a batch that is a multiple of 8 piece w belongs to matrix w mod 8
(one XCD per matrix: the fence-free hand-offs of batch_step.hip depend on it).  No GPU."""
import ctypes as C

import numpy as np
import pytest

D, LA, P, T, TD, AL = range(6)


def plan(B, n, has_y, order=0):
    from volt_amd import _lib
    L = _lib.lib()
    cnt = L.volt_batch_describe(B, n, has_y, order, None, 0)
    assert cnt > 0
    buf = (C.c_int * (4 * cnt))()
    assert L.volt_batch_describe(B, n, has_y, order, buf, cnt) == cnt
    assert L.volt_batch_describe(B, n, has_y, order, buf, cnt - 1) == -2
    return np.array(buf).reshape(cnt, 4)


@pytest.mark.parametrize("B,n,has_y,order", [(8, 4, 1, 0), (3, 5, 1, 0), (16, 7, 1, 1), (5, 6, 0, 0), (1, 1, 1, 0), (2, 2, 1, 0), (64, 32, 1, 0),
                                             (8, 9, 1, 2 << 1), (16, 12, 0, 1 << 1), (8, 32, 1, 2 << 1 | 1),
                                             (64, 32, 1, 1054), (16, 9, 1, 1022), (64, 8, 0, 1054), (24, 12, 1, 1118), (64, 32, 1, -1)])
def test_list_is_topological_and_complete(B, n, has_y, order):
    # order < 16: look-ahead tiles listed (order >> 1) block columns early (the one sanctioned forward wait); 1000 + w: round 6's
    # windowed orders (batch_sched.h) -- what 64 x 4096 runs (order -1 = the step's own choice) -- which only permute a block
    # column's tiles among the matrices: the same pieces, still behind everything they wait for, still matrix w % 8 at index w
    lad = order >> 1 if 0 <= order < 16 else 0
    it = plan(B, n, has_y, order)
    kind, b, row, col = it[:, 0] & 7, it[:, 0] >> 3, it[:, 1], it[:, 2]
    pos = {}
    for w, (k, bb, r, c) in enumerate(zip(kind, b, row, col)):
        key = (int(k), int(bb), int(r), int(c))
        assert key not in pos, f"piece {key} twice"
        pos[key] = w
    for bb in range(B):
        for k in range(n):
            assert (D, bb, k, 0) in pos
            if 1 <= k <= n - 2:
                assert (LA, bb, k, 0) in pos
            for i in range(k + 1, n):
                assert (P, bb, i, k) in pos
        if has_y:
            for i in range(n):
                assert (TD, bb, i, 0) in pos
                for j in range(i):
                    assert (T, bb, i, j) in pos
                for q in range(-(-(i + 1) * 128 // 1024)):
                    assert (AL, bb, i, q) in pos
    extra = sum(-(-(i + 1) * 128 // 1024) for i in range(n)) + n + n * (n - 1) // 2 if has_y else 0
    assert len(pos) == B * (n + max(0, n - 2) + n * (n - 1) // 2 + extra)

    def before(a, c):
        assert pos[a] < pos[c], (a, c)

    for (k, bb, r, c), w in pos.items():
        if k == D and r >= 1:
            before((P, bb, r, r - 1), (k, bb, r, c))                  # L[k, k-1]
            if r >= 2:
                before((LA, bb, r - 1, 0), (k, bb, r, c))             # the look-ahead part of A[k, k]
        elif k == LA:
            if lad == 0:
                before((P, bb, r + 1, r - 1), (k, bb, r, c))          # row k+1 up to column k-1
            else:                                                     # listed early: behind D(max(1, k - lad)), and what it
                before((D, bb, max(1, r - lad), 0), (k, bb, r, c))    # still waits for is at most `lad` block columns on
                assert pos[(P, bb, r + 1, r - 1)] < pos[(D, bb, min(r + 1, n - 1), 0)]
            before((k, bb, r, c), (D, bb, r + 1, 0))                  # and ahead of the diagonal tile that needs it
        elif k == P:
            before((D, bb, c, 0), (k, bb, r, c))                      # W_k
            if c >= 1:
                before((P, bb, r, c - 1), (k, bb, r, c))
                before((P, bb, c, c - 1), (k, bb, r, c))
        elif k == TD:
            before((D, bb, r, 0), (k, bb, r, c))
        elif k == T:
            before((D, bb, r, 0), (k, bb, r, c))                      # W_i
            before((P, bb, r, r - 1), (k, bb, r, c))                  # all of row i of L
            before((TD, bb, c, 0) if r - 1 == c else (T, bb, r - 1, c), (k, bb, r, c))    # Y[j, i-1]
        elif k == AL:
            before((TD, bb, r, 0), (k, bb, r, c))
            for j in range(r):
                before((T, bb, r, j), (k, bb, r, c))
    if B % 8 == 0:
        assert np.array_equal(b % 8, np.arange(len(it)) % 8)          # piece w <-> matrix w (mod 8) <-> XCD w % 8


def test_gate_is_a_function_of_the_shape_only():
    from volt_amd import _lib
    L = _lib.lib()
    # the workspace of a shape the one launch takes carries its piece list; where it does not run nothing is reserved
    grow = lambda B, N: L.volt_mll_workspace_bytes(B, N, 1)
    assert grow(64, 4096) > 64 * (2 * 4096 * 4096 * 4)                # A and Y and more
    assert L.volt_potrf_workspace_bytes(2, 1024) == L.volt_potrf_workspace_bytes(2, 1024)   # deterministic
    assert L.volt_batch_describe(0, 4, 1, 0, None, 0) == -1 and L.volt_batch_describe(4, 4, 1, 16, None, 0) == -1


# ---- the fp64 one-launch step (csrc/batch64_step.hip): no table, the piece is a function of blockIdx -------------------
D64, US64, TD64, T64, LA64 = range(5)


def plan64(B, n, has_y):
    from volt_amd import _lib
    L = _lib.lib()
    cnt = L.volt_batch64_describe(B, n, has_y, None, 0)
    assert cnt > 0
    buf = (C.c_int * (4 * cnt))()
    assert L.volt_batch64_describe(B, n, has_y, buf, cnt) == cnt
    assert L.volt_batch64_describe(B, n, has_y, buf, cnt - 1) == -2
    return np.array(buf).reshape(cnt, 4)


@pytest.mark.parametrize("B,n,has_y", [(1, 32, 0), (1, 32, 1), (8, 8, 1), (3, 5, 1), (2, 2, 1), (5, 3, 0), (16, 4, 1), (1, 1, 1)])
def test_fp64_list_is_topological_and_complete(B, n, has_y):
    it = plan64(B, n, has_y)
    kind, row, col, mat = it.T
    pos = {}
    for w, key in enumerate(zip(kind.tolist(), mat.tolist(), row.tolist(), col.tolist())):
        assert key not in pos, f"piece {key} twice"
        pos[key] = w
    # the matrix is the innermost index: with a batch that is a multiple of 8 a matrix stays on one XCD
    assert (mat == np.arange(len(it)) % B).all()
    for bb in range(B):
        for i in range(n):
            assert (D64, bb, i, i) in pos                              # D(i) carries tile (i, i-1) too
            if 2 <= i <= n - 1:
                assert (LA64, bb, i, i) in pos
            for k in range(i - 1):
                assert (US64, bb, i, k) in pos
            if has_y:
                assert (TD64, bb, i, i) in pos
                for j in range(i):
                    assert (T64, bb, i, j) in pos
    assert len(pos) == B * (n + max(0, n - 2) + (n - 1) * (n - 2) // 2 + (n * (n + 1) // 2 if has_y else 0))

    def row_done(bb, i, k):
        """the piece that publishes rowp[i] = k + 1 (tile (i, k) of L)"""
        return (D64, bb, i, i) if k == i - 1 else (US64, bb, i, k)

    def before(a, c):
        assert pos[a] < pos[c], (a, c)

    for (k, bb, r, c), w in pos.items():
        me = (k, bb, r, c)
        if k == D64 and r >= 1:
            before((D64, bb, r - 1, r - 1), me)                        # the sub-blocks of L[r-1, r-1]
            if r >= 2:
                before((LA64, bb, r, r), me)
                before(row_done(bb, r, r - 2), me)                     # the last block the chased sum of tile (r, r-1) reads
                before(row_done(bb, r - 1, r - 2), me)
        if k == LA64:
            before(row_done(bb, r, r - 2), me)
        if k == US64:
            before((D64, bb, c, c), me)
            if c >= 1:
                before(row_done(bb, r, c - 1), me)
                before(row_done(bb, c, c - 1), me)
        if k == TD64:
            before((D64, bb, r, r), me)
        if k == T64:                                                   # tile (r, c) of the inverse
            before((D64, bb, r, r), me)                                # W_r
            before(row_done(bb, r, r - 1), me)                         # all of row r of L
            before((TD64, bb, c, c), me)
            if r - 1 > c:
                before((T64, bb, r - 1, c), me)
