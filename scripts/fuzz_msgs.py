"""Developer tool: time-boxed random parity run of the message-side kernels against the oracle: toOccupancyGrid /
fromOccupancyGrid (random geometry, layer contents incl. NaN / out-of-range / +-inf, data ranges, moved buffers),
LaserScan -> rays (random beam counts, increments either side of the 0.017 rad decimation threshold, ranges at and
beyond range_min / range_max, NaN / inf returns).  usage: python scripts/fuzz_msgs.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ros_navigation_amd as R  # noqa: E402
import _oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.zeros(1, device="cuda")
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = cells = nrays = inexact = 0
while time.time() < t_end:
    res = float(rng.choice([0.05, 0.1, 0.2, 0.025]))
    lx, ly = float(rng.uniform(2, 20)), float(rng.uniform(2, 20))
    px, py = float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))
    here = dict(res=res, lx=lx, ly=ly, px=px, py=py, fuzz_seed=seed, case=cases)
    e = R.Engine(lx, ly, res, px, py)
    g = O.make_geom(lx, ly, res, px, py)
    layer = rng.choice(np.array([np.nan, 0, 10, 50, 99.5, 100, 150, 180, 254.9, 255, 256, -0.5, -3, 1e9, np.inf, -np.inf], np.float32), e.ncell)
    noise = rng.random(e.ncell) < 0.3
    layer[noise] = rng.uniform(-10, 300, int(noise.sum())).astype(np.float32)
    for l in range(3):
        e.upload(l, layer)
    ref = layer.copy()
    if rng.random() < 0.5:
        target = (px + float(rng.uniform(-0.4, 0.4)) * lx, py + float(rng.uniform(-0.4, 0.4)) * ly)
        ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
        regs = (O.Region * 4)()
        mv = C.c_int(0)
        O.lib().og_move(C.byref(g), ptrs, 1, O.d2(*target), regs, C.byref(mv))
        e.move(*target)
        here["moved"] = target
    for dmin, dmax in ((0.0, 255.0), (-1.0, 100.0), (float(rng.uniform(-5, 50)), float(rng.uniform(60, 300)))):
        want = O.to_occupancy_grid(g, ref, dmin, dmax)
        got = e.to_occupancy_grid(R.capi.LAYER_LASER, dmin, dmax)
        if not np.array_equal(got, want):
            bad = np.flatnonzero(got != want)
            print("MISMATCH to_occupancy", here, dmin, dmax, bad[:5], got[bad[:5]], want[bad[:5]])
            sys.exit(1)
    cells += e.ncell
    if "moved" not in here:
        data = rng.integers(-1, 101, e.ncell).astype(np.int8)
        data[rng.random(e.ncell) < 0.02] = rng.integers(-128, 128, 1).astype(np.int8)[0]
        e.from_occupancy_grid(R.capi.LAYER_RANGE, data)
        got, want = e.download(R.capi.LAYER_RANGE), O.from_occupancy_grid(e.rows, e.cols, data)
        if not (np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])):
            print("MISMATCH from_occupancy", here)
            sys.exit(1)
    # laser scans
    ns, beams = int(rng.integers(1, 12)), int(rng.choice([1, 2, 37, 181, 361, 1081, 2500]))
    inc = np.float32(rng.choice([0.0005, 0.004, 0.0169, 0.017, 0.0171, 0.05, 1.5 * np.pi / max(beams, 1)]))
    scans, ranges = R.synth.laser_scans(ns, beams, lx, ly, seed=int(rng.integers(0, 1 << 30)), angle_increment=inc)
    ranges[rng.random(len(ranges)) < 0.05] = np.float32(rng.choice([np.nan, np.inf, 0.0, -1.0]))
    k = rng.random(len(ranges))
    per_scan_max = np.repeat(scans["range_max"], scans["n_ranges"])[: len(ranges)]
    per_scan_min = np.repeat(scans["range_min"], scans["n_ranges"])[: len(ranges)]
    ranges[k < 0.05] = per_scan_max[k < 0.05]
    ranges[(k > 0.05) & (k < 0.1)] = per_scan_min[(k > 0.05) & (k < 0.1)]
    want = O.scan_to_rays(scans, ranges)
    got = e.scan_to_rays(scans, ranges)
    ok = (len(got) == len(want) and np.array_equal(got["sx"], want["sx"]) and np.array_equal(got["sy"], want["sy"]) and
          np.array_equal(got["clear_end"], want["clear_end"]) and np.allclose(got["ex"], want["ex"], rtol=0, atol=1e-6) and
          np.allclose(got["ey"], want["ey"], rtol=0, atol=1e-6))
    if not ok:
        print("MISMATCH scan_to_rays", here, dict(ns=ns, beams=beams, inc=float(inc)), len(got), len(want))
        sys.exit(1)
    nrays += len(want)
    inexact += int(((got["ex"] != want["ex"]) | (got["ey"] != want["ey"])).sum())
    # the same with full sensor transforms (tilted / rolled / raised mounts, end quaternions of either sign)
    scans_tf, ranges_tf = R.synth.laser_scans_tf(ns, beams, lx, ly, seed=int(rng.integers(0, 1 << 30)), angle_increment=inc,
                                                 tilt=float(rng.choice([0.05, 0.35, 1.2])), planar=float(rng.choice([0.0, 0.25, 1.0])))
    ranges_tf[rng.random(len(ranges_tf)) < 0.05] = np.float32(rng.choice([np.nan, np.inf, 0.0, -1.0]))
    want = O.scan_to_rays_tf(scans_tf, ranges_tf)
    got = e.scan_to_rays_tf(scans_tf, ranges_tf)
    ok = (len(got) == len(want) and np.array_equal(got["sx"], want["sx"]) and np.array_equal(got["sy"], want["sy"]) and
          np.array_equal(got["clear_end"], want["clear_end"]) and np.allclose(got["ex"], want["ex"], rtol=0, atol=1e-6) and
          np.allclose(got["ey"], want["ey"], rtol=0, atol=1e-6))
    if not ok:
        print("MISMATCH scan_to_rays_tf", here, dict(ns=ns, beams=beams, inc=float(inc)), len(got), len(want))
        sys.exit(1)
    nrays += len(want)
    inexact += int(((got["ex"] != want["ex"]) | (got["ey"] != want["ey"])).sum())
    cases += 1
    e.close()
print("msgs fuzz ok: %d maps, %d cells converted, %d rays from scans (%d end points differ by a float32 ulp) in %.0f s, seed %d"
      % (cases, cells, nrays, inexact, budget, seed))
