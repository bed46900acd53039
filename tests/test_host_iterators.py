"""CPU-only: the product's host-side walks (rna_line_cells / rna_circle_cells / rna_submap_cells and the
geometry-only index math of librna.so -- no engine, no GPU) against the oracle's restatement of
grid_map_core's iterators, which the reference's own gtest vectors pin (tests/test_oracle_gridmap.py):
LineIteratorTest.cpp:45-109, SubmapIteratorTest.cpp:28-167, GridMapMathTest.cpp:26-155."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O


@pytest.fixture(scope="module")
def capi():
    import subprocess, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "ros_navigation_amd", "csrc"), "-j4", "-s"])
    from ros_navigation_amd import capi
    return capi


def oracle_cells(fn, *args, width=2):
    cap = 1 << 16
    out = np.zeros(width * cap, np.int32)
    n = fn(*args, out.ctypes.data_as(C.POINTER(C.c_int32)), cap)
    assert n <= cap
    return out[:width * n].reshape(-1, width)[:, :2].copy()


GEOMS = [((8.1, 5.1), 1.0, (0.0, 0.0), (0, 0)),          # LineIteratorTest.cpp's 8 x 5 map
         ((2.0, 2.0), 0.05, (0.0, 0.0), (0, 0)),
         ((6.4, 4.8), 0.05, (1.3, -0.7), (17, 40)),       # shifted and moved (circular buffer)
         ((3.0, 5.0), 0.1, (-2.0, 4.0), (29, 0))]


@pytest.mark.parametrize("length,res,pos,start", GEOMS)
def test_index_math_and_walks_match_the_oracle(capi, length, res, pos, start):
    g = capi.make_geometry(length[0], length[1], res, pos, start)
    og = O.make_geom(length[0], length[1], res, pos[0], pos[1], start)
    assert (g.size[0], g.size[1]) == (og.size[0], og.size[1])
    rng = np.random.default_rng(7)
    half = 0.75 * max(length)
    for _ in range(400):
        x, y = rng.uniform(-half, half, 2) + np.array(pos)
        idx = (C.c_int * 2)()
        inside = O.lib().og_index_from_position(C.byref(og), O.d2(x, y), idx)
        got = capi.geometry_index(g, x, y)
        assert (got is not None) == bool(inside)
        if inside:
            assert got == (idx[0], idx[1])
            p = (C.c_double * 2)()
            O.lib().og_position_from_index(C.byref(og), idx, p)
            assert capi.geometry_position(g, *got) == (p[0], p[1])
    assert capi.geometry_position(g, g.size[0], 0) is None and capi.geometry_position(g, 0, -1) is None
    for _ in range(300):
        sx, sy, ex, ey = rng.uniform(-half, half, 4) + np.array([pos[0], pos[1], pos[0], pos[1]])
        want = oracle_cells(O.lib().og_line_cells, C.byref(og), O.d2(sx, sy), O.d2(ex, ey))
        assert np.array_equal(capi.line_cells(g, sx, sy, ex, ey), want)
    for _ in range(200):
        cx, cy = rng.uniform(-half, half, 2) + np.array(pos)
        r = rng.uniform(0.0, 0.4 * max(length))
        want = oracle_cells(O.lib().og_circle_cells, C.byref(og), O.d2(cx, cy), r)
        assert np.array_equal(capi.circle_cells(g, cx, cy, r), want)
    for _ in range(100):
        tl = (int(rng.integers(0, g.size[0])), int(rng.integers(0, g.size[1])))
        sz = (int(rng.integers(1, g.size[0] + 1)), int(rng.integers(1, g.size[1] + 1)))
        want = oracle_cells(O.lib().og_submap_cells, C.byref(og), O.i2(*tl), O.i2(*sz), width=4)
        assert np.array_equal(capi.submap_cells(g, tl, sz), want)


def test_line_iterator_known_answers_of_the_reference(capi):
    """gmc/test/LineIteratorTest.cpp:45-109 on the product's host walk directly (8 x 5 map, resolution 1, origin)."""
    g = capi.make_geometry(8.0, 5.0, 1.0)
    assert capi.line_cells(g, 0.0, 0.0, 9.0, 6.0).tolist() == [[4, 2], [3, 1], [2, 1], [1, 0], [0, 0]]     # :45-72
    c = capi.line_cells(g, -7.0, -9.0, 8.0, 8.0).tolist()                                                  # :74-99
    assert c[:3] == [[5, 4], [4, 3], [3, 2]] and 3 <= len(c) <= 6
    assert len(capi.line_cells(g, -8.0, 8.0, 8.0, 8.0)) == 0                                               # :101-109
    # malformed rays are dropped whole, as the HIMM kernels drop them
    assert len(capi.line_cells(g, float("inf"), 0.0, 0.0, 0.0)) == 0 and len(capi.line_cells(g, 0.0, 0.0, 1e9, 0.0)) == 0
