"""Generates tests/golden/vfh_golden.npz from the REFERENCE's own vfh.cpp (oracle/_ref, built by
`make -C oracle ref` in the container that has /root/reference).  The fixture is data only:
inputs (range scans, speeds, goals, pinned clock steps) and the reference's outputs.

    python tests/golden/gen_vfh_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402


def param_sets():
    a = O.default_vfh_params()                      # Steerer defaults, mc/src/steerer.cpp:69-121
    b = O.default_vfh_params()                      # custom_description/src/vfh_node.cpp:35-87 style
    b.window_diameter = 60
    b.robot_radius = 300.0
    b.safety_dist_0ms = 100.0
    b.safety_dist_1ms = 100.0
    b.max_turnrate_0ms = 80
    b.max_turnrate_1ms = 40
    b.max_speed_narrow_opening = 50
    b.weight_desired_dir = 5.0
    b.weight_current_dir = 3.0
    return [a, b]


def params_to_array(p):
    return np.array([getattr(p, f) for f, _ in p._fields_], dtype=np.float64)


def main():
    assert O.ref() is not None, "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20261001)
    n_seq, n_step = 24, 20
    out = {}
    for pi, p in enumerate(param_sets()):
        ranges = np.full((n_seq, n_step, 181), 5000.0)
        speed = np.zeros((n_seq, n_step), np.int32)
        gdir = np.zeros((n_seq, n_step), np.float32)
        gdist = np.zeros((n_seq, n_step), np.float32)
        gtol = np.full((n_seq, n_step), 250.0, np.float32)
        dt = np.zeros((n_seq, n_step), np.float64)
        o_hist = np.zeros((n_seq, n_step, 72), np.float32)
        o_origin = np.zeros((n_seq, n_step, 72), np.float32)
        o_picked = np.zeros((n_seq, n_step), np.float32)
        o_speed = np.zeros((n_seq, n_step), np.int32)
        o_turn = np.zeros((n_seq, n_step), np.int32)
        for s in range(n_seq):
            v = O.RefVfh(p)
            mode = s % 4
            for k in range(n_step):
                if mode:
                    nobs = rng.integers(0, 50)
                    idx = rng.integers(0, 181, nobs)
                    ranges[s, k, idx] = rng.uniform(150 if mode == 3 else 350, 3000, nobs)
                if mode == 2:
                    a0 = rng.integers(0, 150)
                    ranges[s, k, a0:a0 + rng.integers(5, 40)] = rng.uniform(400, 1500)
                speed[s, k] = rng.integers(-10, 200)
                gdir[s, k] = rng.uniform(0, 360) if rng.random() < 0.7 else 90.0
                gdist[s, k] = rng.uniform(100, 4000)
                dt[s, k] = [0.2, 0.25, 0.125, 0.5, 0.0625][rng.integers(0, 5)]
                full = np.full(361, 5000.0)
                full[0::2] = ranges[s, k]
                cs, ct = v.update(full, int(speed[s, k]), gdir[s, k], gdist[s, k], gtol[s, k], float(dt[s, k]))
                o_hist[s, k] = v.hist()
                o_origin[s, k] = v.origin_hist()
                o_picked[s, k] = v.picked_angle()
                o_speed[s, k], o_turn[s, k] = cs, ct
        pre = "p%d_" % pi
        out.update({pre + "params": params_to_array(p), pre + "ranges_even": ranges, pre + "speed": speed,
                    pre + "goal_dir": gdir, pre + "goal_dist": gdist, pre + "goal_tol": gtol, pre + "dt": dt,
                    pre + "hist": o_hist, pre + "origin_hist": o_origin, pre + "picked": o_picked,
                    pre + "chosen_speed": o_speed, pre + "chosen_turnrate": o_turn})
    np.savez_compressed(os.path.join(HERE, "vfh_golden.npz"), **out)
    print("wrote vfh_golden.npz:", {k: v.shape for k, v in out.items() if k.startswith("p0_")})


if __name__ == "__main__":
    main()
