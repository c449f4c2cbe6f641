"""The piece list of the one-launch step for ONE long series (csrc/long_sched.h, through volt_long_describe): host logic,
no GPU.  Every tile exactly once, the K ranges of a tile's slices partition its early part, and -- what the launch's
freedom from deadlock rests on -- every piece depends only on pieces BEFORE it in the list (workgroups are dispatched
in grid order)."""
import numpy as np
import pytest

from volt_amd import _lib

D0, SPINE, PANEL, U, T, TDIAG, E_PANEL, E_T, E_U, R = range(10)


def plan(n, first=4, emin=2):
    L = _lib.lib()
    ns, nc = np.zeros(1, np.int32), np.zeros(1, np.int32)
    k = L.volt_long_describe(n, first, emin, None, 0, ns.ctypes.data, nc.ctypes.data)
    items = np.zeros((k, 4), np.int32)
    assert L.volt_long_describe(n, first, emin, items.ctypes.data, k, None, None) == k
    return items, int(ns[0]), int(nc[0])


@pytest.mark.parametrize("n,first,emin", [(9, 4, 2), (12, 4, 2), (16, 3, 2), (32, 4, 2), (32, 2, 2), (20, 6, 2),
                                          (3, 3, 1), (4, 3, 1), (5, 3, 1), (8, 3, 1), (24, 3, 1), (32, 4, 1), (32, 4, 0), (3, 0, -1), (8, 0, -1), (24, 0, -1), (32, 0, -1)])
def test_pieces_are_in_dependency_order(n, first, emin):
    items, nslabs, ncnt = plan(n, first, emin)
    lf, yf, wf, uf = set(), set(), set(), set()      # tiles of L stored, tiles of Y stored, diagonal blocks done, look-ahead tiles parked
    cnt = {}                                          # slices delivered per counter
    ranges = {}                                       # counter -> list of (b0, b1)
    seen = set()
    slabs = set()
    for q, (x, y, z, w) in enumerate(items.tolist()):
        kind, a, b = x & 255, (x >> 8) & 255, (x >> 16) & 255

        def early(tile_blocks):                       # a base piece: its early part is either its own (needs the blocks' inputs) or slices
            if y:
                assert cnt.get(w, 0) == y, f"piece {q}: {cnt.get(w, 0)} of {y} slices out"
                rs = sorted(ranges[w])
                assert rs[0][0] == 0 and rs[-1][1] == tile_blocks and all(p[1] == r[0] for p, r in zip(rs, rs[1:]))
                assert set(range(z, z + y)) <= slabs
                return True
            return False

        is_slice = E_PANEL <= kind <= E_U
        key = (kind if not is_slice else -kind, a, b, y if is_slice else 0)
        assert key not in seen, f"piece {q} twice"
        seen.add(key)
        if kind == D0:
            wf.add(0)
        elif kind == SPINE:
            k, kd = a, a - 1
            if not early(kd - 1) and kd > 1:
                assert (k, kd - 2) in lf and (kd, kd - 2) in lf
            if kd >= 1:
                assert (k, kd - 1) in lf and (kd, kd - 1) in lf
            assert kd in wf
            if k >= 3:
                assert k in uf
            lf.add((k, kd))
            split = q + 1 < len(items) and (items[q + 1][0] & 255) == R
            if split:                                 # split spine: R(k), right behind it, owns the diagonal block
                assert ((items[q + 1][0] >> 8) & 255) == k
            else:
                wf.add(k)
        elif kind == R:
            assert (a, a - 1) in lf and (q == 0 or (items[q - 1][0] & 255, (items[q - 1][0] >> 8) & 255) == (SPINE, a))
            if a >= 3:
                assert a in uf
            wf.add(a)
        elif kind == PANEL:
            i, k = a, b
            if not early(k - 1) and k > 1:
                assert (i, k - 2) in lf and (k, k - 2) in lf
            if k >= 1:
                assert (i, k - 1) in lf and (k, k - 1) in lf
            assert k in wf
            lf.add((i, k))
        elif kind == U:
            assert a >= 3 and early(a - 2)
            uf.add(a)
        elif kind == T:
            i, j = a, b
            if not early(i - j - 1) and i - j > 1:
                assert (i, i - 2) in lf and (i - 2, j) in yf
            assert (i, i - 1) in lf and (i - 1, j) in yf and i in wf
            yf.add((i, j))
        elif kind == TDIAG:
            assert a in wf
            yf.add((a, a))
        else:
            b0, b1 = y & 255, (y >> 8) & 255
            assert 0 <= b0 < b1
            if kind == E_PANEL:
                assert (a, b1 - 1) in lf and (b, b1 - 1) in lf
            elif kind == E_T:
                assert (a, b + b1 - 1) in lf and (b + b1 - 1, b) in yf
            else:
                assert (a, b1 - 1) in lf
            assert z not in slabs and 0 <= z < nslabs and 0 <= w < ncnt
            slabs.add(z)
            cnt[w] = cnt.get(w, 0) + 1
            ranges.setdefault(w, []).append((b0, b1))
    # every tile of the factor and of the inverse exactly once
    assert lf == {(i, k) for i in range(n) for k in range(i)}
    assert yf == {(i, j) for i in range(n) for j in range(i + 1)}
    assert wf == set(range(n))


def test_slices_grow_away_from_the_tile_and_spread_evenly():
    """The last slice of an early part is `first` blocks, the ones before it twice as long each; at 32 block columns no
    block column completes more than ~80 slices (the launch keeps one workgroup per CU: 256 resident)."""
    items, nslabs, ncnt = plan(32, 4)
    per_tile = {}
    for x, y, z, w in items.tolist():
        if E_PANEL <= (x & 255) <= E_U:
            per_tile.setdefault(w, []).append(((y >> 8) & 255) - (y & 255))
    for lens in per_tile.values():
        # in list (= readiness) order: a remainder first, then 4 * 2^i blocks falling to the 4 next to the tile
        assert lens[-1] <= 4 and all(a == 2 * b for a, b in zip(lens[1:], lens[2:])) and (len(lens) < 2 or lens[0] <= 2 * lens[1])
    assert 1500 < nslabs < 2200 and len(items) < 3200
