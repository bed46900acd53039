"""Pins oracle/vfh.c: (a) against committed golden vectors produced by the reference's own
vfh.cpp (tests/golden/gen_vfh_golden.py), (b) live against oracle/_ref when it is built here.
Bit-exact on every output (OriginHist, Hist, picked angle, speed, turnrate)."""
import os

import numpy as np
import pytest

import _oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vfh_golden.npz")


def params_from_array(a):
    p = O.VfhParams()
    for (name, ctype), v in zip(p._fields_, a):
        setattr(p, name, int(v) if "int" in ctype.__name__ else float(v))
    return p


@pytest.mark.parametrize("pi", [0, 1])
def test_oracle_matches_reference_golden(pi):
    z = np.load(GOLD)
    pre = "p%d_" % pi
    p = params_from_array(z[pre + "params"])
    R = z[pre + "ranges_even"]
    n_seq, n_step = R.shape[:2]
    for s in range(n_seq):
        v = O.OracleVfh(p)
        for k in range(n_step):
            full = np.full(361, 5000.0)
            full[0::2] = R[s, k]
            cs, ct = v.update(full, int(z[pre + "speed"][s, k]), z[pre + "goal_dir"][s, k],
                              z[pre + "goal_dist"][s, k], z[pre + "goal_tol"][s, k], float(z[pre + "dt"][s, k]))
            assert cs == z[pre + "chosen_speed"][s, k] and ct == z[pre + "chosen_turnrate"][s, k], (s, k)
            assert v.hist().tobytes() == z[pre + "hist"][s, k].tobytes(), (s, k)
            assert v.origin_hist().tobytes() == z[pre + "origin_hist"][s, k].tobytes(), (s, k)
            assert np.float32(v.picked_angle()).tobytes() == z[pre + "picked"][s, k].tobytes(), (s, k)


@pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (reference sources absent)")
def test_oracle_matches_reference_live_fuzz():
    rng = np.random.default_rng(7)
    p = O.default_vfh_params()
    for s in range(40):
        o, r = O.OracleVfh(p), O.RefVfh(p)
        for k in range(15):
            ranges = np.full(361, 5000.0)
            n = rng.integers(0, 60)
            ranges[rng.integers(0, 181, n) * 2] = rng.uniform(150, 3000, n)
            args = (int(rng.integers(0, 200)), np.float32(rng.uniform(0, 360)), np.float32(rng.uniform(100, 4000)),
                    np.float32(250), [0.2, 0.5, 0.125][rng.integers(0, 3)])
            assert o.update(ranges, *args) == r.update(ranges, *args)
            assert o.hist().tobytes() == r.hist().tobytes()
            assert o.origin_hist().tobytes() == r.origin_hist().tobytes()
            assert o.picked_angle() == r.picked_angle()


@pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (reference sources absent)")
def test_oracle_tables_match_reference():
    L, R = O.lib(), O.ref()
    p = O.default_vfh_params()
    o, r = O.OracleVfh(p), O.RefVfh(p)
    W = p.window_diameter
    T = L.og_vfh_num_tables(o.h)
    assert T == R.refvfh_num_tables(r.h) == 20
    cd = np.ctypeslib.as_array(L.og_vfh_cell_direction(o.h), (W * W,))
    for x in range(W):
        for y in range(W):
            assert cd[x * W + y] == R.refvfh_cell_direction(r.h, x, y) or (x == y == 15)
            for t in (0, 7, 19):
                n = L.og_vfh_cell_sector_count(o.h, t, x, y)
                assert n == R.refvfh_cell_sector_count(r.h, t, x, y)
                lst = L.og_vfh_cell_sector_list(o.h, t, x, y)
                assert [lst[k] for k in range(n)] == [R.refvfh_cell_sector(r.h, t, x, y, k) for k in range(n)]
    for s in range(p.max_speed + 1):
        assert L.og_vfh_min_turning_radius(o.h, s) == R.refvfh_min_turning_radius(r.h, s)


def test_emergency_stop_branch():
    """Obstacle inside the safety radius => histogram all 1, speed 0, spin (vfh.cpp:533-540,1070-1077)."""
    v = O.OracleVfh()
    ranges = np.full(361, 5000.0)
    ranges[180] = 120.0  # straight ahead, closer than robot_radius + safety
    cs, ct = v.update(ranges, 0, np.float32(90), np.float32(2000), np.float32(250), 0.2)
    assert cs == 0 and ct == 40
    assert np.all(v.origin_hist() == 1.0)
