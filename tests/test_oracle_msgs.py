"""Oracle pinning for the message formats either side of the path:
known answers of the reference's own tests, grid_map_ros/test/GridMapRosTest.cpp:116-184."""
import ctypes as C
import math

import numpy as np

import _oracle as O

L = O.lib()


def _move(g, layer, x, y):
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(layer))
    regs = (O.Region * 4)()
    moved = C.c_int(0)
    L.og_move(C.byref(g), ptrs, 1, O.d2(x, y), regs, C.byref(moved))
    return moved.value


# GridMapRosTest.cpp:116-138 (OccupancyGridConversion.withMove)
def test_occupancy_grid_with_move():
    g = O.make_geom(8.0, 5.0, 0.5)
    assert tuple(g.size) == (16, 10)
    layer = np.full(160, 1.0, np.float32)
    occ = O.to_occupancy_grid(g, layer, 0.0, 1.0)
    assert occ[0] == 100 and (occ == 100).all()
    assert _move(g, layer, -1.0, -1.0) == 1
    occ2 = O.to_occupancy_grid(g, layer, 0.0, 1.0)
    assert occ2[0] == -1                       # cell (0, 0) of the message is unobserved now
    assert (occ2 == -1).sum() == 160 - 14 * 8  # 2 rows and 2 columns of cells were dropped
    assert set(np.unique(occ2).tolist()) == {-1, 100}


# GridMapRosTest.cpp:140-184 (OccupancyGridConversion.roundTrip): width 50, height 100, values in
# [-1, 100], fromOccupancyGrid then toOccupancyGrid(-1, 100) reproduces every cell
def test_occupancy_grid_round_trip():
    rng = np.random.default_rng(5)
    width, height = 50, 100
    data = rng.integers(-1, 101, width * height).astype(np.int8)
    layer = O.from_occupancy_grid(width, height, data)
    assert np.isnan(layer[::-1][data == -1]).all()
    g = O.make_geom(0.1 * width, 0.1 * height, 0.1, 3.0 + 0.05 * width, 6.0 + 0.05 * height)
    assert tuple(g.size) == (width, height)
    back = O.to_occupancy_grid(g, layer, -1.0, 100.0)
    assert (back == data).all()
    # origin convention (GridMapRosConverter.cpp:212-214,261-263): position - length/2
    assert math.isclose(g.pos[0] - 0.5 * g.len[0], 3.0) and math.isclose(g.pos[1] - 0.5 * g.len[1], 6.0)


# MapProvider::publishMap scaling (map_provider.cpp:116,211): 0..255 -> 0..100, truncation, clamp
def test_occupancy_grid_scaling_of_himm_values():
    g = O.make_geom(0.2, 0.2, 0.05)
    layer = np.array([np.nan, 0, 10, 30, 150, 180, 255, 300, -5, 127.5, 2.55, 2.54, 252.45, 1e9, -0.0, 179.9], np.float32)
    occ = O.to_occupancy_grid(g, layer, 0.0, 255.0)[::-1]   # message order is the reverse of the layer
    want = []
    for v in layer:
        f = np.float32((np.float32(v) - np.float32(0)) / np.float32(255))
        want.append(-1 if (math.isnan(f) or f < 0) else int(np.float32(min(max(np.float32(0), f), np.float32(1))) * np.float32(100)))
    assert occ.tolist() == want
    assert occ.tolist()[:9] == [-1, 0, 3, 11, 58, 70, 100, 100, -1]


# Steerer::pubHist (steerer.cpp:201-220): unpinned by the reference, checked against the formula
def test_hist_msg_packing():
    rng = np.random.default_rng(2)
    origin = (rng.random(72) * 5e7).astype(np.float32)
    hist = rng.integers(0, 2, 72).astype(np.float32)
    x, y, yb, th = O.hist_msg(hist, origin, 5)
    assert x.tolist() == [5 * i for i in range(36)] and th == (2000, 4000)
    assert yb.tolist() == hist[:36].astype(np.int64).tolist()
    assert y.tolist() == [int(v) & 0xFFFF for v in origin[:36]]


# Nav::taileredPlan (nav_node.cpp:192-204)
def test_tailor_plan():
    plan = np.stack([np.arange(13, dtype=np.float64), -np.arange(13, dtype=np.float64)], 1)
    out = O.tailor_plan(plan, 5)
    assert out[:, 0].tolist() == [12.0, 10.0, 5.0, 0.0]
    assert O.tailor_plan(plan[:1], 5)[:, 0].tolist() == [0.0]
    assert len(O.tailor_plan(plan[:0], 5)) == 0
