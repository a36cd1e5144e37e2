"""The fold of the walkers' two-box tree into four-wide nodes on the 16-bit grid (csrc/host/fold_nodes.h, rt_types.h GpuNode4Q), run on
host memory through the test hooks — no GPU.  Random trees with leaves at every depth and empty children: the fold must keep every leaf
exactly once, give every record a grid box that holds the float box it came from with a cell to spare, halve the depth, and refuse what
does not fit."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAF, EMPTY = 0x80000000, 0xFFFFFFFF


@pytest.fixture(scope="module")
def hooks():
    path = os.path.join(ROOT, "raytracing-course-hw_amd", "librtamd_testhooks.so")
    if not os.path.exists(path):
        pytest.skip("test hooks not built")
    L = C.CDLL(path)
    L.rtt_fold_nodes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.rtt_fold_nodes.restype = C.c_int
    return L


def random_tree(rng, n_leaves, lo, hi, empty_root_child=False):
    """Two-box nodes as uint32 words [n, 16]: lo0 child0 hi0 cnt0 lo1 child1 hi1 cnt1.  Returns (words, depth, leaf words)."""
    nodes, leaves = [], []

    def build(blo, bhi, count, depth):
        """returns (child word, depth below)"""
        if count == 1 or (count <= 3 and rng.random() < 0.3):
            w = LEAF | len(leaves)
            leaves.append(w)
            return w, depth
        idx = len(nodes)
        nodes.append(None)
        axis = int(rng.integers(0, 3))
        cut = blo[axis] + (bhi[axis] - blo[axis]) * rng.uniform(0.3, 0.7)
        nl = int(rng.integers(1, count))
        l_hi, r_lo = bhi.copy(), blo.copy()
        l_hi[axis] = cut + 0.05 * (bhi[axis] - blo[axis]); r_lo[axis] = cut - 0.05 * (bhi[axis] - blo[axis])  # overlapping children
        c0, d0 = build(blo.copy(), l_hi, nl, depth + 1)
        c1, d1 = build(r_lo, bhi.copy(), count - nl, depth + 1)
        rec = np.zeros(16, np.uint32)
        rec[0:3] = blo.astype(np.float32).view(np.uint32); rec[3] = c0
        rec[4:7] = l_hi.astype(np.float32).view(np.uint32)
        rec[8:11] = r_lo.astype(np.float32).view(np.uint32); rec[11] = c1
        rec[12:15] = bhi.astype(np.float32).view(np.uint32)
        nodes[idx] = rec
        return idx, max(d0, d1)

    root, depth = build(np.array(lo, float), np.array(hi, float), n_leaves, 1)
    if root & LEAF:                                         # a one-leaf tree: the root wraps it, the other child is empty
        rec = np.zeros(16, np.uint32)
        rec[0:3] = np.array(lo, np.float32).view(np.uint32); rec[3] = root
        rec[4:7] = np.array(hi, np.float32).view(np.uint32)
        rec[8:11] = np.float32(3.0e38).view(np.uint32); rec[11] = EMPTY
        rec[12:15] = np.float32(3.0e38).view(np.uint32)
        nodes.append(rec); depth = 1
    words = np.stack(nodes)
    if empty_root_child and not (root & LEAF):
        words[0, 11] = EMPTY                                 # drop the root's second subtree (its nodes stay in the array, unreachable)
    return np.ascontiguousarray(words), depth, leaves


def fold(hooks, words, lo, hi):
    out = np.zeros((len(words) + 2, 16), np.uint32)
    depth = C.c_uint32(0)
    grid = np.zeros(9, np.float32)
    box = np.array(list(lo) + list(hi), np.float32)
    n = hooks.rtt_fold_nodes(words.ctypes.data, len(words), box.ctypes.data, out.ctypes.data, len(out), C.byref(depth), grid.ctypes.data)
    return n, out[:max(n, 0)], depth.value, grid


def f32(w):
    return np.array(w, np.uint32).view(np.float32).astype(np.float64)


@pytest.mark.parametrize("n_leaves,seed", [(1, 0), (2, 1), (3, 2), (7, 3), (64, 4), (1000, 5), (20000, 6)])
def test_fold_keeps_every_leaf_and_every_box(hooks, n_leaves, seed):
    rng = np.random.default_rng(seed)
    lo, hi = (-3.0, 0.5, -40.0), (5.0, 0.75, 10.0)
    words, depth2, leaves = random_tree(rng, n_leaves, lo, hi, empty_root_child=(seed % 2 == 1 and n_leaves > 3))
    n, wide, depth4, grid = fold(hooks, words, lo, hi)
    assert n >= 1
    g_lo, step = grid[0:3].astype(np.float64), grid[3:6].astype(np.float64)
    seen = []

    def expected_entries(b):
        """(lo, hi, child) of the children of two-box node b's children, in record order"""
        out = []
        for (l, h, c) in ((words[b, 0:3], words[b, 4:7], int(words[b, 3])), (words[b, 8:11], words[b, 12:15], int(words[b, 11]))):
            if c == EMPTY:
                continue
            if c & LEAF:
                out.append((l, h, c))
            else:
                for (l2, h2, c2) in ((words[c, 0:3], words[c, 4:7], int(words[c, 3])), (words[c, 8:11], words[c, 12:15], int(words[c, 11]))):
                    if c2 != EMPTY:
                        out.append((l2, h2, c2))
        return out

    reached_depth = 0
    todo = [(0, 0, 1)]                                       # (wide node, two-box node, level)
    while todo:
        w, b, level = todo.pop()
        reached_depth = max(reached_depth, level)
        exp = expected_entries(b)
        for k in range(4):
            rec = wide[w, 4 * k: 4 * k + 4]
            if k >= len(exp):
                assert int(rec[3]) == EMPTY and not rec[0:3].any()
                continue
            l, h, c = exp[k]
            cell_lo, cell_hi = (rec[0:3] & 0xFFFF).astype(np.float64), (rec[0:3] >> 16).astype(np.float64)
            box_lo, box_hi = g_lo + cell_lo * step, g_lo + cell_hi * step
            assert (box_lo <= f32(l) - 0.99 * step).all() and (box_hi >= f32(h) + 0.99 * step).all()      # a cell to spare on either side
            assert (box_lo >= f32(l) - 2.01 * step).all() and (box_hi <= f32(h) + 2.01 * step).all()      # and no more than two
            if c & LEAF:
                assert int(rec[3]) == c
                seen.append(c)
            else:
                assert not (int(rec[3]) & LEAF) and 0 < int(rec[3]) < n
                todo.append((int(rec[3]), c, level + 1))
    reachable = [l for l in leaves] if not (words[0, 11] == EMPTY and len(words) > 1 and not (int(words[0, 3]) & LEAF and len(leaves) == 1)) else None
    if reachable is not None:
        assert sorted(seen) == sorted(reachable)                                                    # every leaf exactly once
    else:
        assert len(seen) == len(set(seen)) and set(seen) <= set(leaves)
    assert depth4 == reached_depth and depth4 <= (depth2 + 1) // 2 + (1 if depth2 == 1 else 0)
    print(f"{n_leaves} leaves: {len(words)} two-box nodes, depth {depth2} -> {n} wide nodes, depth {depth4}")


def test_fold_refuses_what_does_not_fit(hooks):
    rng = np.random.default_rng(9)
    lo, hi = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    words, _, _ = random_tree(rng, 50, lo, hi)
    assert fold(hooks, words, lo, hi)[0] > 0
    assert fold(hooks, words, (0.2, 0.0, 0.0), hi)[0] == -1          # the grid does not hold the tree's boxes
    bad = words.copy(); bad[0, 3] = len(words) + 5                   # a child index beyond the array
    assert fold(hooks, bad, lo, hi)[0] == -1
    loop = words.copy(); loop[1, 3] = 0                              # a cycle: node 1 points back at the root
    assert fold(hooks, loop, lo, hi)[0] == -1
