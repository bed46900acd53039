"""Pins the oracle's grid_map_core restatement (oracle/gridmath.c) to the reference's own gtest
known answers.  gmc/ = /root/reference/grid_map-master/grid_map_core ; each test cites the gtest
case it re-expresses (inputs and expected values only -- no reference code).
"""
import ctypes as C
import math
import sys

import numpy as np
import pytest

import _oracle as O

EPS = sys.float_info.epsilon
L = O.lib()


def idx_from_pos(g, x, y):
    out = O.i2(0, 0)
    ok = L.og_index_from_position(C.byref(g), O.d2(x, y), out)
    return bool(ok), (out[0], out[1])


def pos_from_idx(g, i, j):
    out = O.d2(0, 0)
    ok = L.og_position_from_index(C.byref(g), O.i2(i, j), out)
    return bool(ok), (out[0], out[1])


def deq(a, b):
    """EXPECT_DOUBLE_EQ: within 4 ULPs."""
    return abs(a - b) <= 4 * EPS * max(abs(a), abs(b), 1e-300) or a == b


# gmc/test/GridMapMathTest.cpp:26-51
def test_position_from_index_simple():
    g = O.raw_geom((3.0, 2.0), (-1.0, 2.0), 1.0, (3, 2))
    for idx, exp in (((0, 0), (1.0, 0.5)), ((1, 0), (0.0, 0.5)), ((1, 1), (0.0, -0.5)), ((2, 1), (-1.0, -0.5))):
        ok, p = pos_from_idx(g, *idx)
        assert ok and deq(p[0], exp[0] - 1.0) and deq(p[1], exp[1] + 2.0)
    assert not pos_from_idx(g, 3, 1)[0]


# gmc/test/GridMapMathTest.cpp:53-83
def test_position_from_index_circular_buffer():
    g = O.raw_geom((0.5, 0.4), (-0.1, 13.4), 0.1, (5, 4), (3, 1))
    for idx, exp in (((3, 1), (0.2, 0.15)), ((4, 2), (0.1, 0.05)), ((2, 0), (-0.2, -0.15)),
                     ((0, 0), (0.0, -0.15)), ((4, 3), (0.1, -0.05))):
        ok, p = pos_from_idx(g, *idx)
        assert ok and deq(p[0], exp[0] + -0.1) and deq(p[1], exp[1] + 13.4)
    assert not pos_from_idx(g, 5, 3)[0]


# gmc/test/GridMapMathTest.cpp:85-115
def test_index_from_position_simple():
    mp = (-12.4, -7.1)
    g = O.raw_geom((3.0, 2.0), mp, 1.0, (3, 2))
    for pos, exp in (((1.0, 0.5), (0, 0)), ((-1.0, -0.5), (2, 1)), ((0.6, 0.1), (0, 0)),
                     ((0.4, -0.1), (1, 1)), ((0.4, 0.1), (1, 0))):
        ok, i = idx_from_pos(g, pos[0] + mp[0], pos[1] + mp[1])
        assert ok and i == exp
    assert not idx_from_pos(g, 4.0 + mp[0], 0.5 + mp[1])[0]


# gmc/test/GridMapMathTest.cpp:117-138
def test_index_from_position_edge_cases():
    g = O.raw_geom((3.0, 2.0), (0.0, 0.0), 1.0, (3, 2))
    assert idx_from_pos(g, 0.0, EPS) == (True, (1, 0))
    assert idx_from_pos(g, 0.5 - EPS, -EPS) == (True, (1, 1))
    assert idx_from_pos(g, -0.5 - EPS, -EPS) == (True, (2, 1))
    assert not idx_from_pos(g, -1.5, 1.0)[0]


# gmc/test/GridMapMathTest.cpp:140-156
def test_index_from_position_circular_buffer():
    mp = (0.4, -0.9)
    g = O.raw_geom((0.5, 0.4), mp, 0.1, (5, 4), (3, 1))
    assert idx_from_pos(g, 0.2 + mp[0], 0.15 + mp[1]) == (True, (3, 1))
    assert idx_from_pos(g, 0.03 + mp[0], -0.17 + mp[1]) == (True, (0, 0))


def within(pos, length, mp):
    return bool(L.og_position_within_map(O.d2(*pos), O.d2(*length), O.d2(*mp)))


# gmc/test/GridMapMathTest.cpp:158-194
def test_position_within_map():
    ln, mp = (50.0, 25.0), (11.4, 0.0)
    for p in ((0, 0), (5, 5), (20, 10), (20, -10), (-20, 10), (-20, -10)):
        assert within((p[0] + mp[0], p[1] + mp[1]), ln, mp)
    ln, mp = (10.0, 5.0), (-3.0, 145.2)
    for p in ((5.5, 0.0), (-5.5, 0.0), (-5.5, 3.0), (-5.5, -3.0), (3.0, 3.0)):
        assert not within((p[0] + mp[0], p[1] + mp[1]), ln, mp)
    ln, mp = (2.0, 3.0), (0.0, 0.0)
    assert not within((1.0, -1.5), ln, mp)
    assert not within((-1.0, 1.5), ln, mp)
    assert not within((1.0 + EPS, 1.0), ln, mp)
    assert within(((2.0 + EPS) / 2.0, 1.0), ln, mp)
    assert not within((0.5, -1.5 - (2.0 * EPS)), ln, mp)
    assert within((-0.5, (3.0 + EPS) / 2.0), ln, mp)


# gmc/test/GridMapMathTest.cpp:195-237
def test_index_and_position_shift():
    def ishift(v, res):
        out = O.i2(0, 0)
        L.og_index_shift_from_position_shift(O.d2(*v), res, out)
        return out[0], out[1]
    assert ishift((0.0, 0.0), 1.0) == (0, 0)
    assert ishift((0.35, -0.45), 1.0) == (0, 0)
    assert ishift((0.55, -0.45), 1.0) == (-1, 0)
    assert ishift((-1.3, -2.65), 1.0) == (1, 3)
    assert ishift((-0.4, 0.09), 0.2) == (2, 0)

    def pshift(v, res):
        out = O.d2(0, 0)
        L.og_position_shift_from_index_shift(O.i2(*v), res, out)
        return out[0], out[1]
    a = pshift((0, 0), 0.3)
    assert deq(a[0], 0.0) and deq(a[1], 0.0)
    a = pshift((1, -1), 0.3)
    assert deq(a[0], -0.3) and deq(a[1], 0.3)
    a = pshift((2, 1), 0.3)
    assert deq(a[0], -0.6) and deq(a[1], -0.3)


# gmc/test/GridMapMathTest.cpp:239-290
def test_index_range_and_wrap():
    bs = O.i2(10, 15)
    rng = lambda i, j: bool(L.og_index_within_range(O.i2(i, j), bs))
    assert rng(0, 0) and rng(9, 14)
    assert not rng(10, 5) and not rng(5, 300) and not rng(-1, 0) and not rng(0, -300)
    for i, e in ((0, 0), (1, 1), (-1, 9), (9, 9), (10, 0), (11, 1), (35, 5), (-9, 1), (-19, 1)):
        assert L.og_wrap_index(i, 10) == e


# gmc/test/GridMapMathTest.cpp:292-402
@pytest.mark.parametrize("mp,cases", [
    ((0.0, 0.0), [((0.0, 0.0), (0.0, 0.0), None), ((15.0, 5.0), (15.0, 5.0), "ge"), ((-15.0, -5.0), (-15.0, -5.0), "le"),
                  ((16.0, 6.0), (15.0, 5.0), "ge"), ((-16.0, -6.0), (-15.0, -5.0), "le"),
                  ((1e6, 1e6), (15.0, 5.0), "ge"), ((-1e6, -1e6), (-15.0, -5.0), "le")]),
    ((1.0, 2.0), [((0.0, 0.0), (0.0, 0.0), None), ((16.0, 7.0), (16.0, 7.0), "ge"), ((-14.0, -3.0), (-14.0, -3.0), "le"),
                  ((17.0, 8.0), (16.0, 7.0), "ge"), ((-15.0, -4.0), (-14.0, -3.0), "le"),
                  ((1e6, 1e6), (16.0, 7.0), "ge"), ((-1e6, -1e6), (-14.0, -3.0), "le")]),
])
def test_limit_position_to_range(mp, cases):
    eps = 11.0 * EPS
    for pin, exp, side in cases:
        p = O.d2(*pin)
        L.og_limit_position_to_range(p, O.d2(30.0, 10.0), O.d2(*mp))
        for a in range(2):
            if side is None:
                assert deq(p[a], exp[a])
            else:
                assert abs(p[a] - exp[a]) <= abs(pin[a]) * eps
                assert (exp[a] >= p[a]) if side == "ge" else (exp[a] <= p[a])


def submap_info(g, rp, rl):
    o = O.SubmapInfo()
    ok = L.og_submap_information(C.byref(g), O.d2(*rp), O.d2(*rl), C.byref(o))
    return bool(ok), o


# gmc/test/GridMapMathTest.cpp:404-437, 439-472, 474-527
def test_submap_information_plain():
    g = O.raw_geom((5.0, 4.0), (0.0, 0.0), 1.0, (5, 4))
    ok, o = submap_info(g, (0.0, 0.5), (0.9, 2.9))
    assert ok and tuple(o.top_left) == (2, 0) and tuple(o.size) == (1, 3)
    assert deq(o.pos[0], 0.0) and deq(o.pos[1], 0.5) and deq(o.len[0], 1.0) and deq(o.len[1], 3.0)
    assert tuple(o.requested_index) == (0, 1)
    ok, o = submap_info(g, (-1.0, -0.5), (0.0, 0.0))
    assert ok and tuple(o.top_left) == (3, 2) and tuple(o.size) == (1, 1)
    assert deq(o.pos[0], -1.0) and deq(o.pos[1], -0.5) and deq(o.len[0], 1.0) and deq(o.len[1], 1.0)
    assert tuple(o.requested_index) == (0, 0)
    ok, o = submap_info(g, (2.0, 1.5), (2.9, 2.9))
    assert ok and tuple(o.top_left) == (0, 0) and tuple(o.size) == (2, 2)
    assert deq(o.pos[0], 1.5) and deq(o.pos[1], 1.0) and deq(o.len[0], 2.0) and deq(o.len[1], 2.0)
    assert tuple(o.requested_index) == (0, 0)
    ok, o = submap_info(g, (0.0, 0.0), (1e6, 1e6))
    assert ok and tuple(o.top_left) == (0, 0) and tuple(o.size) == (5, 4)
    assert deq(o.pos[0], 0.0) and deq(o.pos[1], 0.0) and deq(o.len[0], 5.0) and deq(o.len[1], 4.0)
    assert o.requested_index[0] == 2 and 1 <= o.requested_index[1] <= 2


# gmc/test/GridMapMathTest.cpp:529-597
def test_submap_information_circular_buffer():
    g = O.raw_geom((5.0, 4.0), (0.0, 0.0), 1.0, (5, 4), (2, 1))
    ok, o = submap_info(g, (0.0, 0.5), (0.9, 2.9))
    assert ok and tuple(o.top_left) == (4, 1) and tuple(o.size) == (1, 3)
    assert deq(o.pos[0], 0.0) and deq(o.pos[1], 0.5) and deq(o.len[0], 1.0) and deq(o.len[1], 3.0)
    assert tuple(o.requested_index) == (0, 1)
    ok, o = submap_info(g, (2.0, 1.5), (2.9, 2.9))
    assert ok and tuple(o.top_left) == (2, 1) and tuple(o.size) == (2, 2)
    assert deq(o.pos[0], 1.5) and deq(o.pos[1], 1.0) and deq(o.len[0], 2.0) and deq(o.len[1], 2.0)
    assert tuple(o.requested_index) == (0, 0)
    ok, o = submap_info(g, (0.0, 0.0), (1e6, 1e6))
    assert ok and tuple(o.top_left) == (2, 1) and tuple(o.size) == (5, 4)
    assert deq(o.len[0], 5.0) and deq(o.len[1], 4.0)
    assert o.requested_index[0] == 2 and 1 <= o.requested_index[1] <= 2


# gmc/test/GridMapMathTest.cpp:599-654
def test_submap_information_debug_cases():
    g = O.raw_geom((4.98, 4.98), (-4.98, -5.76), 0.06, (83, 83), (0, 13))
    ok, o = submap_info(g, (-7.44, -3.42), (0.12, 0.12))
    assert ok and tuple(o.size) == (2, 3) and deq(o.len[0], 0.12) and deq(o.len[1], 0.18)
    g = O.raw_geom((4.98, 4.98), (2.46, -25.26), 0.06, (83, 83), (42, 6))
    ok, o = submap_info(g, (0.24, -26.82), (0.624614, 0.462276))
    assert ok and o.size[0] > 0 and o.size[1] > 0 and o.len[0] > 0 and o.len[1] > 0


def regions(si, ss, bs, start=(0, 0)):
    out = (O.Region * 4)()
    n = L.og_buffer_regions_for_submap(O.i2(*si), O.i2(*ss), O.i2(*bs), O.i2(*start), out)
    return n, [(r.quadrant, tuple(r.index), tuple(r.size)) for r in out[:max(n, 0)]]


TL, TR, BL, BR = 1, 2, 3, 4


# gmc/test/GridMapMathTest.cpp:656-761
def test_buffer_regions_for_submap():
    assert regions((0, 0), (0, 0), (5, 4)) == (1, [(TL, (0, 0), (0, 0))])
    assert regions((0, 0), (0, 7), (5, 4))[0] == -1
    assert regions((0, 0), (6, 7), (5, 4))[0] == -1
    assert regions((1, 2), (3, 2), (5, 4)) == (1, [(TL, (1, 2), (3, 2))])
    st = (3, 1)
    assert regions((3, 1), (2, 3), (5, 4), st) == (1, [(TL, (3, 1), (2, 3))])
    assert regions((4, 1), (2, 3), (5, 4), st) == (2, [(TL, (4, 1), (1, 3)), (BL, (0, 1), (1, 3))])
    assert regions((1, 0), (2, 1), (5, 4), st) == (1, [(BR, (1, 0), (2, 1))])
    assert regions((3, 1), (5, 4), (5, 4), st) == (4, [(TL, (3, 1), (2, 3)), (TR, (3, 0), (2, 1)),
                                                       (BL, (0, 1), (3, 3)), (BR, (0, 0), (3, 1))])


# gmc/test/GridMapMathTest.cpp:763-852
def test_increment_index():
    idx, bs, st = O.i2(0, 0), O.i2(4, 3), O.i2(0, 0)
    seq = []
    while L.og_increment_index(idx, bs, st):
        seq.append((idx[0], idx[1]))
    assert seq[:4] == [(0, 1), (0, 2), (1, 0), (1, 1)]
    assert seq[9] == (3, 1) and seq[10] == (3, 2) and len(seq) == 11
    idx, st = O.i2(2, 1), O.i2(2, 1)
    seq = []
    while L.og_increment_index(idx, bs, st):
        seq.append((idx[0], idx[1]))
    assert seq == [(2, 2), (2, 0), (3, 1), (3, 2), (3, 0), (0, 1), (0, 2), (0, 0), (1, 1), (1, 2), (1, 0)]


# gmc/test/GridMapMathTest.cpp:854-943
def test_increment_index_for_submap():
    def run(tl, ss, bs, st, sub0):
        sub, idx = O.i2(*sub0), O.i2(0, 0)
        ok = L.og_increment_index_for_submap(sub, idx, O.i2(*tl), O.i2(*ss), O.i2(*bs), O.i2(*st))
        return bool(ok), (sub[0], sub[1]), (idx[0], idx[1])
    tl, ss, bs = (3, 1), (2, 4), (8, 5)
    assert run(tl, ss, bs, (0, 0), (0, 0)) == (True, (0, 1), (3, 2))
    assert run(tl, ss, bs, (0, 0), (0, 1)) == (True, (0, 2), (3, 3))
    assert run(tl, ss, bs, (0, 0), (0, 2)) == (True, (0, 3), (3, 4))
    assert run(tl, ss, bs, (0, 0), (0, 3)) == (True, (1, 0), (4, 1))
    assert run(tl, ss, bs, (0, 0), (1, 2)) == (True, (1, 3), (4, 4))
    assert not run(tl, ss, bs, (0, 0), (1, 3))[0]
    assert not run(tl, ss, bs, (0, 0), (2, 0))[0]
    tl, st = (6, 3), (3, 2)
    assert run(tl, ss, bs, st, (0, 0)) == (True, (0, 1), (6, 4))
    assert run(tl, ss, bs, st, (0, 1)) == (True, (0, 2), (6, 0))
    assert run(tl, ss, bs, st, (0, 2)) == (True, (0, 3), (6, 1))
    assert run(tl, ss, bs, st, (0, 3)) == (True, (1, 0), (7, 3))
    assert run(tl, ss, bs, st, (1, 2)) == (True, (1, 3), (7, 1))
    assert not run(tl, ss, bs, st, (1, 3))[0]
    assert not run(tl, ss, bs, st, (2, 0))[0]


# gmc/test/GridMapMathTest.cpp:945-953
def test_index_from_linear_index():
    def f(lin, size, rm):
        out = O.i2(0, 0)
        L.og_index_from_linear(lin, O.i2(*size), rm, out)
        return out[0], out[1]
    assert f(0, (8, 5), 0) == (0, 0) and f(1, (8, 5), 0) == (1, 0) and f(1, (8, 5), 1) == (0, 1)
    assert f(2, (8, 5), 0) == (2, 0) and f(8, (8, 5), 0) == (0, 1) and f(39, (8, 5), 0) == (7, 4)


def line(g, s, e):
    cells = (C.c_int * 4096)()
    n = L.og_line_cells(C.byref(g), O.d2(*s), O.d2(*e), cells, 2048)
    return [(cells[2 * k], cells[2 * k + 1]) for k in range(n)]


# gmc/test/LineIteratorTest.cpp:45-109 (:20-43 is inconsistent as vendored -- SURVEY.md section 4)
def test_line_iterator():
    g = O.make_geom(8.0, 5.0, 1.0)
    assert line(g, (0.0, 0.0), (9.0, 6.0)) == [(4, 2), (3, 1), (2, 1), (1, 0), (0, 0)]
    cells = line(g, (-7.0, -9.0), (8.0, 8.0))
    # the gtest checks the first three cells, then three more ++ must reach isPastEnd(): 3 <= n <= 6
    assert cells[:3] == [(5, 4), (4, 3), (3, 2)] and 3 <= len(cells) <= 6
    assert cells == [(5, 4), (4, 3), (3, 2), (2, 1), (1, 0)]  # hand-derived in SURVEY.md section 4
    assert line(g, (-8.0, 8.0), (8.0, 8.0)) == []


# gmc/test/SubmapIteratorTest.cpp:28-89 and :91-167
def test_submap_iterator():
    g = O.make_geom(8.1, 5.1, 1.0)
    assert tuple(g.size) == (8, 5)
    out = (C.c_int * 400)()
    n = L.og_submap_cells(C.byref(g), O.i2(3, 1), O.i2(3, 2), out, 100)
    got = [tuple(out[4 * k:4 * k + 4]) for k in range(n)]
    assert got == [(3, 1, 0, 0), (3, 2, 0, 1), (4, 1, 1, 0), (4, 2, 1, 1), (5, 1, 2, 0), (5, 2, 2, 1)]
    # circular buffer: map.move(Position(-3,-2)) -> startIndex (3,2)
    layer = np.zeros(40, np.float32)
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(layer))
    regs = (O.Region * 4)()
    moved = C.c_int(0)
    L.og_move(C.byref(g), ptrs, 1, O.d2(-3.0, -2.0), regs, C.byref(moved))
    assert tuple(g.start) == (3, 2)
    n = L.og_submap_cells(C.byref(g), O.i2(6, 3), O.i2(2, 4), out, 100)
    got = [tuple(out[4 * k:4 * k + 4]) for k in range(n)]
    assert got == [(6, 3, 0, 0), (6, 4, 0, 1), (6, 0, 0, 2), (6, 1, 0, 3),
                   (7, 3, 1, 0), (7, 4, 1, 1), (7, 0, 1, 2), (7, 1, 1, 3)]


# gmc/test/GridMapTest.cpp:57-85
def test_gridmap_move():
    g = O.make_geom(8.1, 5.1, 1.0)
    layer = np.zeros(40, np.float32)
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(layer))
    regs = (O.Region * 4)()
    moved = C.c_int(0)
    n = L.og_move(C.byref(g), ptrs, 1, O.d2(-3.0, -2.0), regs, C.byref(moved))
    assert tuple(g.start) == (3, 2) and moved.value == 1
    m = layer.reshape(5, 8).T  # m[i, j], column-major storage
    assert math.isnan(m[0, 0]) and not math.isnan(m[3, 2]) and math.isnan(m[2, 2])
    assert math.isnan(m[3, 1]) and not math.isnan(m[7, 4])
    assert n == 2
    assert (tuple(regs[0].index), tuple(regs[0].size)) == ((0, 0), (3, 5))
    assert (tuple(regs[1].index), tuple(regs[1].size)) == ((0, 0), (8, 2))
    # every cell: valid iff unwrapped row >= 3 - ... i.e. rows 3..7 and cols 2..4 kept
    for i in range(8):
        for j in range(5):
            assert math.isnan(m[i, j]) == (i < 3 or j < 2)


# gmc/test/GridMapIteratorTest.cpp:27-63 : 40 cells, linear index i + j*rows
def test_gridmap_iterator_linear_order():
    size = O.i2(8, 5)
    seen = set()
    for lin in range(40):
        out = O.i2(0, 0)
        L.og_index_from_linear(lin, size, 0, out)
        assert out[0] + out[1] * 8 == lin
        seen.add((out[0], out[1]))
    assert len(seen) == 40


# gmc/test/GridMapTest.cpp:173-195 (atPosition nearest): index lookup only
def test_at_position_nearest():
    g = O.make_geom(3.0, 3.0, 1.0)
    assert idx_from_pos(g, 1.35, -0.4) == (True, (0, 1))
    assert idx_from_pos(g, -0.3, 0.0) == (True, (1, 1))


def far_edge_positions(n_geoms=400, seed=9):
    """(geometry, position) pairs within a few ulp of the map's far edge, where ((p - len/2) - pos) / res rounds to
    -size although checkIfPositionWithinMap's strict `<` still holds"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_geoms):
        res = float(rng.choice([0.05, 0.1, 0.2, 0.025]))
        lx, ly = float(rng.uniform(3, 30)), float(rng.uniform(3, 30))
        px, py = (float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))) if rng.random() < 0.5 else (0.0, 0.0)
        g = O.make_geom(lx, ly, res, px, py)
        edge = (g.pos[0] - 0.5 * g.len[0], g.pos[1] - 0.5 * g.len[1])
        for a in range(2):
            v = edge[a]
            for _ in range(6):
                p = [float(rng.uniform(g.pos[0] - 0.4 * g.len[0], g.pos[0] + 0.4 * g.len[0])),
                     float(rng.uniform(g.pos[1] - 0.4 * g.len[1], g.pos[1] + 0.4 * g.len[1]))]
                p[a] = v
                out.append((g, tuple(p)))
                v = float(np.nextafter(v, np.inf))
    return out


def test_index_from_position_never_returns_an_index_past_the_map():
    """The reference's getIndexFromPosition can return index == size for a position within rounding of the far edge
    (unmoved buffer) and then indexes its matrices out of bounds; oracle and kernels define that position as outside
    (gridmath.c og_index_from_position).  Here: whenever the lookup succeeds the index is in range, and the defined
    case does occur among the probes."""
    hit = inside_but_rejected = 0
    for g, p in far_edge_positions():
        idx = O.i2(-7, -7)
        ok = O.lib().og_index_from_position(C.byref(g), O.d2(*p), idx)
        within = O.lib().og_position_within_map(O.d2(*p), g.len, g.pos)
        if ok:
            assert 0 <= idx[0] < g.size[0] and 0 <= idx[1] < g.size[1], (p, idx[0], idx[1])
            hit += 1
        elif within:
            inside_but_rejected += 1
    assert hit > 100 and inside_but_rejected > 0
