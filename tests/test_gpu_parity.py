"""GPU parity tests: the HIP path (through the C ABI in include/rna.h) against the CPU oracle on the
same seeded inputs, and against the committed golden vectors generated from the reference.
Bit-exact for cells/indices/paths/integers and for the VFH histogram floats; RRT waypoints (double
positions built from device atan2/cos/sin) within 1e-9 m."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vfh_golden.npz")


@pytest.fixture(scope="module")
def R():
    import ros_navigation_amd as R
    R.capi.lib()  # fails loudly when librna.so is missing -- there is no fallback
    return R


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same_f32(a, b):
    """bitwise equality, all NaNs treated as equal"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(bits(a)[~np.isnan(a)], bits(b)[~np.isnan(b)])


# ------------------------------------------------------------------------------------------------
# container
# ------------------------------------------------------------------------------------------------
def test_layers_geometry_roundtrip(R):
    e = R.Engine(10.0, 7.5, 0.05, 1.25, -2.5)
    g = O.make_geom(10.0, 7.5, 0.05, 1.25, -2.5)
    assert (e.rows, e.cols) == (g.size[0], g.size[1]) == (200, 150)
    assert np.all(np.isnan(e.download(R.capi.LAYER_MASTER)))  # setGeometry -> NaN
    rng = np.random.default_rng(0)
    a = rng.normal(size=e.ncell).astype(np.float32)
    e.upload(R.capi.LAYER_RANGE, a)
    assert np.array_equal(e.download(R.capi.LAYER_RANGE), a)
    for _ in range(200):
        x, y = rng.uniform(-5, 8), rng.uniform(-8, 3)
        out = O.i2(0, 0)
        ok = O.lib().og_index_from_position(C.byref(g), O.d2(x, y), out)
        got = e.get_index(x, y)
        assert (got is not None) == bool(ok)
        if ok:
            assert got == (out[0], out[1])
            p = O.d2(0, 0)
            O.lib().og_position_from_index(C.byref(g), out, p)
            assert e.get_position(*got) == (p[0], p[1])
    e.close()


def test_move_matches_oracle(R):
    e = R.Engine(8.1, 5.1, 1.0)
    g = O.make_geom(8.1, 5.1, 1.0)
    a = np.arange(40, dtype=np.float32)
    for l in range(3):
        e.upload(l, a)
    ref = a.copy()
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
    regs = (O.Region * 4)()
    mv = C.c_int(0)
    for target in ((-3.0, -2.0), (-2.0, 1.0), (4.4, 0.4), (40.0, 0.0)):
        O.lib().og_move(C.byref(g), ptrs, 1, O.d2(*target), regs, C.byref(mv))
        assert e.move(*target) == bool(mv.value)
        gg = e.geometry()
        assert tuple(gg.start_index) == tuple(g.start) and tuple(gg.position) == tuple(g.pos)
        assert same_f32(e.download(R.capi.LAYER_LASER), ref)
    e.close()


# ------------------------------------------------------------------------------------------------
# HIMM
# ------------------------------------------------------------------------------------------------
def random_rays(rng, n, half, hit=0.7, outside=0.2):
    r = np.zeros(n, O.RAY_DTYPE)
    ox, oy = rng.uniform(-half, half, n), rng.uniform(-half, half, n)
    th, ln = rng.uniform(-np.pi, np.pi, n), rng.uniform(0.0, 1.5 * half, n)
    far = rng.random(n) < outside
    ox[far] *= 2.5
    r["sx"], r["sy"] = ox, oy
    r["ex"], r["ey"] = ox + ln * np.cos(th), oy + ln * np.sin(th)
    r["clear_end"] = (rng.random(n) >= hit).astype(np.int32)
    return r


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 50), (2, 3000), (3, 20000)])
def test_himm_matches_oracle(R, seed, n):
    rng = np.random.default_rng(seed)
    e = R.Engine(6.4, 4.8, 0.05)  # 128 x 96
    g = O.make_geom(6.4, 4.8, 0.05)
    init = rng.choice(np.array([np.nan, 0, 10, 50, 150, 160, 170, 180, 7.5, -3, 1e3], np.float32), e.ncell)
    e.upload(R.capi.LAYER_LASER, init)
    rays = random_rays(rng, n, 3.0)
    # many rays ending in the same few cells (a wall seen by consecutive beams): ordering of marks
    k = max(1, n // 4)
    rays["ex"][:k] = 1.0 + rng.integers(0, 3, k) * 0.05
    rays["ey"][:k] = 0.5
    rays["clear_end"][:k] = 0
    ref = init.copy()
    for lo in range(0, n, 7000):  # several batches in sequence
        chunk = rays[lo:lo + 7000]
        O.himm_update(g, ref, chunk)
        e.himm_update(R.capi.LAYER_LASER, chunk.view(R.capi.RAY_DTYPE))
    assert same_f32(e.download(R.capi.LAYER_LASER), ref)
    e.close()


def test_himm_zero_length_and_outside_rays(R):
    e = R.Engine(2.0, 2.0, 0.05)
    g = O.make_geom(2.0, 2.0, 0.05)
    rays = np.zeros(6, O.RAY_DTYPE)
    rays[0] = (0.3, 0.3, 0.3, 0.3, 0, 0)      # zero length inside: one cell cleared then marked
    rays[1] = (5.0, 5.0, 5.0, 5.0, 0, 0)      # zero length outside
    rays[2] = (-8.0, 8.0, 8.0, 8.0, 1, 0)     # misses the map
    rays[3] = (0.0, 0.0, 3.0, 0.0, 0, 0)      # end outside: clipped, no mark
    rays[4] = (3.0, 3.0, 0.0, 0.0, 0, 0)      # start outside
    rays[5] = (0.999, 0.999, -0.999, -0.999, 1, 0)
    ref = np.full(1600, np.nan, np.float32)
    O.himm_update(g, ref, rays)
    e.himm_update(R.capi.LAYER_LASER, rays.view(R.capi.RAY_DTYPE))
    assert same_f32(e.download(R.capi.LAYER_LASER), ref)
    e.close()


def test_himm_malformed_rays_are_dropped_not_spun_on(R):
    """A non-finite coordinate makes the reference's clipping march (LineIterator.cpp:92-104) spin forever, a start
    10^9 m away makes it spin for minutes.  Defined in oracle/himm.c and himm.hip alike: the ray is dropped whole; its
    neighbours in the batch are applied as usual, and the call returns."""
    e = R.Engine(2.0, 2.0, 0.05)
    g = O.make_geom(2.0, 2.0, 0.05)
    rays = np.zeros(9, O.RAY_DTYPE)
    rays[0] = (0.3, 0.3, -0.4, 0.2, 0, 0)
    rays[1] = (np.inf, 0.0, 0.0, 0.0, 0, 0)
    rays[2] = (0.0, 0.0, 0.0, -np.inf, 0, 0)
    rays[3] = (np.nan, 0.1, 0.2, 0.2, 0, 0)
    rays[4] = (0.1, 0.1, np.nan, np.nan, 1, 0)
    rays[5] = (1e9, 0.0, 0.0, 0.0, 0, 0)        # would be 2e10 steps of the march
    rays[6] = (0.0, 0.0, 0.0, 1e12, 1, 0)
    rays[7] = (-0.5, 0.6, 0.7, -0.1, 0, 0)
    rays[8] = (0.0, 0.0, 5e4, 0.0, 1, 0)        # long but legal (10^6 cells): clipped and cleared
    ref = np.full(1600, 40.0, np.float32)
    O.himm_update(g, ref, rays)
    e.fill(R.capi.LAYER_LASER, 40.0)
    e.himm_update(R.capi.LAYER_LASER, rays.view(R.capi.RAY_DTYPE))
    assert same_f32(e.download(R.capi.LAYER_LASER), ref)
    assert (ref != 40.0).sum() > 40          # the well-formed rays did their work
    e.close()


def test_compose_after_master_was_written_directly(R):
    """compose mode 0 copies only the tiles HIMM flagged -- unless master was written behind its back (upload, fill,
    fromOccupancyGrid, HIMM on the master layer): the reference's `master = laser` (map_provider.cpp:221) replaces
    such content on the next update, so the engine falls back to the whole-layer copy once."""
    e = R.Engine(12.8, 12.8, 0.05)
    g = O.make_geom(12.8, 12.8, 0.05)
    rng = np.random.default_rng(8)
    laser = np.full(e.ncell, np.nan, np.float32)
    rays = random_rays(rng, 300, 3.0, outside=0.0)
    O.himm_update(g, laser, rays)
    e.update_map(rays.view(R.capi.RAY_DTYPE), compose_mode=0)
    junk = rng.integers(0, 7, e.ncell).astype(np.float32) * 30.0
    e.upload(R.capi.LAYER_MASTER, junk)                       # e.g. GridMap::set("master", ...)
    rays = random_rays(rng, 50, 1.0, outside=0.0)             # touches a few tiles only
    O.himm_update(g, laser, rays)
    e.update_map(rays.view(R.capi.RAY_DTYPE), compose_mode=0)
    assert same_f32(e.download(R.capi.LAYER_MASTER), laser)
    _, nbr = O.astar_masks(laser, e.rows, e.cols)
    assert np.array_equal(e.nbr_mask(), nbr)
    # HIMM on the range layer does not flag laser tiles: the following compose leaves master alone
    e.himm_update(R.capi.LAYER_RANGE, random_rays(rng, 50, 2.0, outside=0.0).view(R.capi.RAY_DTYPE))
    e.compose_master(0)
    assert same_f32(e.download(R.capi.LAYER_MASTER), laser)
    assert not e.last_dirty_tiles().any()
    e.close()


def test_update_map_compose_modes(R):
    rng = np.random.default_rng(5)
    for mode in (0, 1):
        e = R.Engine(12.8, 12.8, 0.05)  # 256 x 256 -> 4 x 4 tiles
        g = O.make_geom(12.8, 12.8, 0.05)
        laser = np.full(e.ncell, np.nan, np.float32)
        for step in range(3):
            rays = random_rays(rng, 400, 2.0 + step, outside=0.0)
            O.himm_update(g, laser, rays)
            e.update_map(rays.view(R.capi.RAY_DTYPE), compose_mode=mode)
            assert same_f32(e.download(R.capi.LAYER_LASER), laser)
            assert same_f32(e.download(R.capi.LAYER_MASTER), laser)  # master = laser (map_provider.cpp:221)
            _, nbr = O.astar_masks(laser, e.rows, e.cols)
            assert np.array_equal(e.nbr_mask(), nbr)
        e.close()


# ------------------------------------------------------------------------------------------------
# VFH
# ------------------------------------------------------------------------------------------------
def check_vfh_vs_oracle(R, e, g, master, poses, steps, params=None):
    n = len(poses)
    oracles = [O.OracleVfh(params) for _ in range(n)]
    rng = np.random.default_rng(99)
    for s in range(steps):
        out, origin, hist = e.vfh_step(poses)
        for k in range(n):
            p = poses[k]
            cs, ct = oracles[k].step_pose(g, master, p["x"], p["y"], p["yaw"], int(p["current_speed"]),
                                          p["goal_direction"], p["goal_distance"], p["goal_tolerance"], float(p["dt"]))
            assert (out["chosen_speed"][k], out["chosen_turnrate"][k]) == (cs, ct), (s, k)
            assert bits(origin[k]).tobytes() == bits(oracles[k].origin_hist()).tobytes(), (s, k)
            assert bits(hist[k]).tobytes() == bits(oracles[k].hist()).tobytes(), (s, k)
            assert np.float32(out["picked_angle"][k]) == np.float32(oracles[k].picked_angle()), (s, k)
        # robots move a little and change their state between steps
        poses["x"] += rng.uniform(-0.05, 0.05, n)
        poses["y"] += rng.uniform(-0.05, 0.05, n)
        poses["yaw"] += rng.uniform(-0.2, 0.2, n)
        poses["current_speed"] = out["chosen_speed"]
        poses["goal_direction"] = rng.uniform(0, 360, n).astype(np.float32)


def test_vfh_step_matches_oracle_on_sparse_map(R):
    e = R.Engine(20.0, 20.0, 0.05)  # 400 x 400
    g = O.make_geom(20.0, 20.0, 0.05)
    master = R.synth.occupancy_sparse(e.rows, e.cols, seed=1)
    e.upload(R.capi.LAYER_MASTER, master)
    poses = R.synth.poses(96, 20.0, 20.0, seed=1)
    e.vfh_init(len(poses))
    check_vfh_vs_oracle(R, e, g, master, poses, steps=6)
    e.close()


def test_vfh_step_dense_walls_edges_and_w60(R):
    e = R.Engine(12.0, 12.0, 0.05)
    g = O.make_geom(12.0, 12.0, 0.05)
    master = R.synth.obstacles_rect(e.rows, e.cols, density=0.12, seed=7, side=(2, 12))
    e.upload(R.capi.LAYER_MASTER, master)
    poses = R.synth.poses(64, 12.0, 12.0, seed=3, margin=0.2, speed=60)   # windows clipped at the map edge
    poses["x"][0], poses["y"][0] = 30.0, 0.0                                # robot outside the map
    e.vfh_init(len(poses))
    check_vfh_vs_oracle(R, e, g, master, poses.copy(), steps=5)
    p = O.default_vfh_params()
    p.window_diameter, p.robot_radius, p.safety_dist_0ms, p.safety_dist_1ms = 60, 300.0, 100.0, 100.0
    p.max_turnrate_0ms, p.max_turnrate_1ms = 80, 40
    rp = R.capi.default_vfh_params()
    for f, _ in rp._fields_:
        setattr(rp, f, getattr(p, f))
    e.vfh_init(len(poses), rp)
    check_vfh_vs_oracle(R, e, g, master, poses.copy(), steps=4, params=p)
    e.close()


@pytest.mark.parametrize("res,density,sector_angle,moved", [
    (0.025, 0.45, 5, False),    # 61 x 61 window, 64-column thread grid; ~1 600 obstacle cells in a window: the list overflows
    (0.025, 0.10, 10, True),    # 36 sectors (less than a wavefront), on a moved buffer
    (0.01, 0.20, 5, False),     # 151 x 151 window: rows longer than the workgroup, the cell-by-cell walk
    (0.1, 0.30, 15, True),      # 16 x 16 window, 24 sectors, moved buffer
])
def test_vfh_step_other_resolutions_and_sector_counts(R, res, density, sector_angle, moved):
    """vfh_step_kernel's paths the Steerer's own numbers (5 cm cells: a 31 x 31 window, 72 sectors) do not reach: the window
    walk's 64- and 128-column thread grids, the obstacle list's overflow (more than 1 024 obstacle cells in a window -> the
    window is walked cell by cell), windows with rows longer than the workgroup, histograms with fewer sectors than a
    wavefront has lanes, each also on a moved buffer.  getSubmap + getRangesFromSubmap (gmc/src/GridMap.cpp:287-339,
    mc/src/steerer.cpp:147-191) and Update_VFH against the oracle, bit for bit."""
    L = 6.0 if res < 0.1 else 12.0
    e = R.Engine(L, L, res)
    g = O.make_geom(L, L, res)
    master = R.synth.obstacles_rect(e.rows, e.cols, density=density, seed=13, side=(2, 9))
    e.upload(R.capi.LAYER_MASTER, master)
    ref = master.copy()
    dx, dy = 0.0, 0.0
    if moved:
        dx, dy = 0.73, -0.41
        ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
        regs = (O.Region * 4)()
        mv = C.c_int(0)
        O.lib().og_move(C.byref(g), ptrs, 1, O.d2(dx, dy), regs, C.byref(mv))
        assert e.move(dx, dy) and tuple(e.geometry().start_index) == tuple(g.start) != (0, 0)
        e.upload(R.capi.LAYER_MASTER, ref)     # (the strips the move cleared stay cleared in both)
    p = O.default_vfh_params()
    p.sector_angle = sector_angle
    rp = R.capi.default_vfh_params()
    for f, _ in rp._fields_:
        setattr(rp, f, getattr(p, f))
    poses = R.synth.poses(40, L, L, seed=17, margin=0.3)
    poses["x"] += dx
    poses["y"] += dy
    e.vfh_init(len(poses), rp)
    check_vfh_vs_oracle(R, e, g, ref, poses, steps=3, params=p)
    e.close()


@pytest.mark.parametrize("pi", [0, 1])
def test_vfh_update_matches_reference_golden(R, pi):
    """Update_VFH on caller-provided scans against vectors produced by the reference's own vfh.cpp."""
    z = np.load(GOLD)
    pre = "p%d_" % pi
    rp = R.capi.default_vfh_params()
    for (name, ctype), v in zip(rp._fields_, z[pre + "params"]):
        setattr(rp, name, int(v) if "int" in ctype.__name__ else float(v))
    Rg = z[pre + "ranges_even"]
    n_seq, n_step = Rg.shape[:2]
    e = R.Engine(4.0, 4.0, 0.05)
    e.vfh_init(n_seq, rp)
    for k in range(n_step):
        ranges = np.full((n_seq, 361, 2), 5000.0)
        ranges[:, 0::2, 0] = Rg[:, k]
        poses = np.zeros(n_seq, R.capi.POSE_DTYPE)
        poses["dt"] = z[pre + "dt"][:, k]
        poses["current_speed"] = z[pre + "speed"][:, k]
        poses["goal_direction"] = z[pre + "goal_dir"][:, k]
        poses["goal_distance"] = z[pre + "goal_dist"][:, k]
        poses["goal_tolerance"] = z[pre + "goal_tol"][:, k]
        out, origin, hist = e.vfh_update(ranges, poses)
        assert np.array_equal(out["chosen_speed"], z[pre + "chosen_speed"][:, k]), k
        assert np.array_equal(out["chosen_turnrate"], z[pre + "chosen_turnrate"][:, k]), k
        assert bits(origin).tobytes() == bits(z[pre + "origin_hist"][:, k]).tobytes(), k
        assert bits(hist).tobytes() == bits(z[pre + "hist"][:, k]).tobytes(), k
        assert bits(out["picked_angle"]).tobytes() == bits(z[pre + "picked"][:, k]).tobytes(), k
    e.close()


# ------------------------------------------------------------------------------------------------
# grid A*
# ------------------------------------------------------------------------------------------------
def oracle_pool(fn, n):
    """fn(k) for k < n on every host core (the C oracle releases the GIL)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(32, len(os.sched_getaffinity(0))))) as ex:
        return list(ex.map(fn, range(n)))


def check_rrt_query(res, paths, q, k, g, master, tol=0.2):
    """One RRT query of a batch against oracle/rrt.c, BOTH formulations of extendTree's steering step
    (/root/reference/move_control/src/rrt_planner.cpp:43-51):
      steer = 1  the kernel's own (near + 0.4 (dx, dy) / sqrt(dx^2 + dy^2), IEEE operations only): status, tree size, samples,
                 path length and every way point bit for bit;
      steer = 0  the REFERENCE's (a = atan2(dy, dx); near + 0.4 (cos a, sin a), this libc's libm): the same counts exactly and
                 every way point within 1e-9 m -- the device-side check against the reference's formulation (the one round 5
                 dropped).  A last-bit tie can in principle part the two trees (scripts/fuzz_rrt.py: 4 in 1.9e5 queries, none
                 on any map the tests use), so a test that meets one names it instead of loosening this."""
    args = dict(tol=tol, seed=int(q["seed"][k]), max_samples=int(q["max_samples"][k]))
    ores, opath = O.rrt_plan(g, master, tuple(q["start"][k]), tuple(q["target"][k]), steer=1, **args)
    got = (res["status"][k], res["tree_size"][k], res["samples"][k], res["path_len"][k])
    assert got == (ores.status, ores.tree_size, ores.samples, ores.path_len), (k, "steer=1")
    assert np.array_equal(paths[k, :ores.path_len], opath), (k, "steer=1")
    rres, rpath = O.rrt_plan(g, master, tuple(q["start"][k]), tuple(q["target"][k]), steer=0, **args)
    assert got == (rres.status, rres.tree_size, rres.samples, rres.path_len), (k, "steer=0: the reference's atan2 / cos / sin")
    if rres.path_len:
        assert np.abs(paths[k, :rres.path_len] - rpath).max() < 1e-9, (k, "steer=0 way points")
    return ores.status


def check_astar(R, e, master, queries, max_len, settled_counts=True, **cfg):
    if cfg:
        e.astar_configure(**cfg)
    res, paths = e.astar(queries, max_len)
    settled = e.astar_settled(len(queries)) if settled_counts else None
    _, nbr = O.astar_masks(master, e.rows, e.cols)
    assert np.array_equal(e.nbr_mask(), nbr)
    gw = np.empty(e.ncell, np.int32)
    total_settled = 0
    for k, q in enumerate(queries):
        ores, opath, _ = O.astar_query(nbr, e.rows, e.cols, q["start"], q["goal"], g_work=gw)
        assert res["status"][k] == ores.status, (k, q)
        if ores.status == 0:
            assert res["cost"][k] == ores.cost and res["path_len"][k] == ores.path_len, (k, q)
            assert np.array_equal(paths[k, :ores.path_len], opath), (k, q)
            assert res["expanded"][k] >= ores.settled
            if settled is not None:
                assert settled[k] == ores.settled, (k, q)   # the whole settled g field agrees with the oracle
            total_settled += ores.settled
    return res, total_settled


@pytest.mark.parametrize("rows,cols,density,seed", [(64, 64, 0.25, 1), (200, 120, 0.3, 2), (512, 512, 0.3, 3)])
def test_astar_paths_bit_identical(R, rows, cols, density, seed):
    e = R.Engine(rows * 0.05, cols * 0.05, 0.05)
    assert (e.rows, e.cols) == (rows, cols)
    master = R.synth.obstacles_rect(rows, cols, density=density, seed=seed, side=(2, max(4, rows // 10)))
    master[np.random.default_rng(seed).random(rows * cols) < 0.03] = np.nan  # unknown cells are free
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.astar_queries(40, master, rows, cols, seed=seed)
    rng = np.random.default_rng(seed)
    q["start"][:6] = rng.integers(0, rows * cols, 6)     # arbitrary cells: blocked / disconnected cases
    q["goal"][6] = q["start"][6]                          # trivial query
    for bw in (2828, 8000, 50000):
        if bw != 2828:
            e.astar_job_counters(reset=True)
        res, _ = check_astar(R, e, master, q, rows * cols, bucket_width=bw)
        # what the kernel counts about itself (rna_astar_job_counters: the bench's work_inflation figures) is consistent with the
        # batch's results: one search per query that got as far as searching, every job in `rounds` (jobs per wavefront, rounded up
        # per query), a touched tile per page, rows of 64 cells in `expanded`
        c = e.astar_job_counters()
        searched = int((res["buckets"] > 0).sum())   # (an invalid query or a walled-in start / goal is answered before the search starts)
        assert c["searches"] == searched, (c, searched)
        assert c["jobs_noop"] + c["tiles_touched"] <= c["jobs"] and c["sticky_turns"] <= c["jobs"]
        assert c["rows_written"] * 64 == int(res["expanded"].astype(np.int64).sum())
        waves = 16    # (a batch of <= 32... queries or depth 1 runs sixteen wavefronts per query; the pipelined kernel eight)
        jobs_lo = int(np.maximum(res["rounds"].astype(np.int64) - 1, 0).sum()) * 8
        jobs_hi = int(res["rounds"].astype(np.int64).sum()) * waves
        assert jobs_lo <= c["jobs"] <= jobs_hi, (jobs_lo, c["jobs"], jobs_hi)
        assert c["buckets"] == int(res["buckets"].astype(np.int64).sum())
    e.close()


def test_astar_refuses_maps_beyond_65536_tiles(R):
    """Tile numbers are 16 bits in the search's queue entries: a map of more than 65 536 tiles of 64 x 16 cells
    (8192 x 8192 cells) is refused loudly (RNA_EINVAL), the engine stays usable for everything else.  (The cell-granular
    frontier kernel that served such maps in rounds 1-2 did not handle moved maps and was removed.)"""
    e = R.Engine(8256 * 0.05, 8192 * 0.05, 0.05)          # 129 x 512 tiles
    q = np.zeros(1, R.capi.ASTAR_QUERY_DTYPE)
    q["goal"][0] = 5
    with pytest.raises(R.capi.RnaError) as err:
        e.astar(q, 64)
    assert "65 536 tiles" in str(err.value)
    assert e.get_index(0.0, 0.0) is not None
    e.close()


def test_astar_cost_range_of_the_field_word(R):
    """The tile kernel's field word is 2^30 - g.  A one-cell-wide serpentine (corridors along i, joined alternately at
    either end) makes paths of more than a million cells: a goal half way down still comes out exact (cost 5.6e8, a
    path of 560 000 cells, reported with status 3 because it does not fit the path buffers); a goal whose cost would pass 2^30 ends with status 4 instead of a wrong answer."""
    rows, cols = 1536, 1500
    e = R.Engine(rows * 0.05, cols * 0.05, 0.05)
    m = np.zeros((cols, rows), np.float32)          # m[j, i]
    m[1::2, :] = 180.0                              # walls between the corridors j = 0, 2, 4, ...
    for k, j in enumerate(range(1, cols, 2)):       # a gap at alternating ends
        m[j, rows - 1 if k % 2 == 0 else 0] = 0.0
    e.upload(R.capi.LAYER_MASTER, m.reshape(-1))
    ncorr = (cols + 1) // 2

    def cell(c, i):                                  # corridor c, position i along it
        return (2 * c) * rows + i

    def along(c, i):                                 # cells from the start to (c, i), following the serpentine
        inside = i if c % 2 == 0 else rows - 1 - i
        return c * (rows + 1) + inside               # every turn adds the gap cell
    start = cell(0, 0)
    mid_c, far_c = 365, ncorr - 1
    q = np.zeros(2, R.capi.ASTAR_QUERY_DTYPE)
    q["start"] = start
    q["goal"] = [cell(mid_c, 700), cell(far_c, 5)]
    e.astar_configure(max_queries=2)
    res, paths = e.astar(q, 700000)
    steps_mid = along(mid_c, 700)
    # (status 3: longer than the path staging of a stage; cost and length are still those of the exact field)
    assert res["status"][0] == 3 and res["cost"][0] == 1000 * steps_mid and res["path_len"][0] == steps_mid + 1
    assert 1000 * along(far_c, 5) > 2 ** 30 and res["status"][1] == 4
    e.close()


def test_astar_moved_map_regression_case(R):
    """A case scripts/fuzz_astar.py found (seed 61): 72 x 39 map after GridMap::move, bucket width 2828, 41 queries.
    The optimal path of query 22 steps diagonally from the corner cell of one tile into the corner cell of the tile
    diagonally below it; the wake-up test for that step ran a DPP wave shift inside a short-circuit `||`, i.e. with the
    lanes whose straight step had already succeeded switched off, read nothing from them and left the diagonal tile
    asleep (cost 95624 instead of 95038)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import replay_astar
    assert replay_astar.replay(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "astar_moved_map_case2120.npz"), verbose=False) == []


def test_astar_page_pool(R):
    """The tile kernel hands search pages out on first touch.  (a) Reuse: the same engine serves batches whose
    searches cover different parts of the map, back to back and through every pipeline stage -- the lazy reset must
    return every page to 'unreached'.  (b) A share per query that is too small costs a second pass, never an answer:
    every query still matches the oracle (status 5 is internal since round 3), and the engine goes on with the full
    share afterwards."""
    e = R.Engine(320 * 0.05, 256 * 0.05, 0.05)
    master = R.synth.obstacles_rect(e.rows, e.cols, density=0.25, seed=21, side=(3, 24))
    e.upload(R.capi.LAYER_MASTER, master)
    e.astar_pipeline_depth(2)
    for seed in (1, 2, 3, 4, 5):
        q = R.synth.astar_queries(24, master, e.rows, e.cols, seed=seed)
        check_astar(R, e, master, q, e.ncell, max_queries=32)
    # (b) 80 tiles in the map, 6 pages per query
    e.astar_page_cap(6)
    q = R.synth.astar_queries(48, master, e.rows, e.cols, seed=9)
    res, paths = e.astar(q, e.ncell)
    _, nbr = O.astar_masks(master, e.rows, e.cols)
    outgrew = 0
    for k in range(len(q)):
        ores, opath, _ = O.astar_query(nbr, e.rows, e.cols, q["start"][k], q["goal"][k])
        assert res["status"][k] == ores.status, (k, res[k])
        if ores.status == 0:
            assert res["cost"][k] == ores.cost and np.array_equal(paths[k, :ores.path_len], opath)
        outgrew += ores.settled > 6 * 1024
    assert outgrew > 8, outgrew     # more than one second pass' worth
    e.astar_page_cap(0)
    check_astar(R, e, master, q, e.ncell, max_queries=64)
    e.close()


def test_astar_searches_that_outgrow_their_pages_are_retried(R):
    """When the stages do not fit HBM the engine halves the pages a query may take (the bench configuration runs with
    half a map's worth).  A search that needs more -- a goal that cannot be reached floods its whole component, the
    reference answers "no path" -- must not fail for that: it ends the first pass with status 5 and is searched again in
    one of the stage's eight retry slots, which hold a page per tile.  Here: 80 tiles, 6 pages per query, batches of at
    most 8 queries, among them a goal inside a closed box (free inside, so it is not rejected up front): every answer
    equals the oracle's, none is status 5; also through a pipelined stage and after the slots have been used before."""
    e = R.Engine(320 * 0.05, 256 * 0.05, 0.05)
    master = R.synth.obstacles_rect(e.rows, e.cols, density=0.25, seed=21, side=(3, 24))
    m2 = master.reshape(e.cols, e.rows).copy()     # [j][i]
    m2[100:113, 150:163] = 0.0
    m2[100, 150:163] = m2[112, 150:163] = 180.0    # a closed box, 11 x 11 free cells inside
    m2[100:113, 150] = m2[100:113, 162] = 180.0
    master = m2.reshape(-1).copy()
    e.upload(R.capi.LAYER_MASTER, master)
    e.astar_page_cap(6)
    _, nbr = O.astar_masks(master, e.rows, e.cols)
    inside = 106 * e.rows + 156
    for depth in (1, 3):
        e.astar_pipeline_depth(depth)
        e.astar_configure(max_queries=8)
        for seed in (9, 10, 11):
            q = R.synth.astar_queries(8, master, e.rows, e.cols, seed=seed)
            q["goal"][3] = inside                          # unreachable, not walled in cell by cell
            q["start"][5], q["goal"][5] = inside, inside + 3 * e.rows + 2   # a short search inside the box
            res, paths = e.astar(q, e.ncell)
            assert e.astar_effective_config()[1] == 6
            settled = e.astar_settled(len(q))     # (a retried query is counted in the retry slot that served it)
            needed_retry = 0
            for k in range(len(q)):
                ores, opath, _ = O.astar_query(nbr, e.rows, e.cols, q["start"][k], q["goal"][k])
                assert res["status"][k] == ores.status, (depth, seed, k, res[k])
                if ores.status == 0:
                    assert res["cost"][k] == ores.cost and res["path_len"][k] == ores.path_len
                    assert np.array_equal(paths[k, :ores.path_len], opath)
                    assert settled[k] == ores.settled, (depth, seed, k, settled[k], ores.settled)
                needed_retry += ores.settled > 6 * 1024
            assert res["status"][3] == 1 and needed_retry >= 1
    # more searches of one batch than there are retry slots: the second pass runs as often as it takes, in the order the
    # searches ended (a list the first pass writes -- a scan of the results for status 5 would race with its own answers)
    e.astar_page_cap(2)
    for depth in (1, 3):
        e.astar_pipeline_depth(depth)
        e.astar_configure(max_queries=40)
        q = R.synth.astar_queries(40, master, e.rows, e.cols, seed=12)
        res, paths = e.astar(q, e.ncell)
        settled = e.astar_settled(len(q))
        many = gone = 0
        for k in range(len(q)):
            ores, opath, _ = O.astar_query(nbr, e.rows, e.cols, q["start"][k], q["goal"][k])
            assert res["status"][k] == ores.status, (depth, k, res[k])
            if ores.status == 0:
                assert res["cost"][k] == ores.cost and res["path_len"][k] == ores.path_len
                assert np.array_equal(paths[k, :ores.path_len], opath)
                assert settled[k] == ores.settled or settled[k] == -1, (depth, k, settled[k], ores.settled)
                gone += settled[k] == -1      # (-1: searched again in a pass whose slot has been reused since)
            many += ores.settled > 2 * 1024
        assert many > 8 and gone > 0, (many, gone)
    e.close()


def test_bench_configuration_answers_an_unreachable_goal(R):
    """The bench's configuration (4096 x 4096, 256 queries per batch, thirteen stages here, sixteen in bench.py) does not fit HBM with a page
    per tile and query: the engine runs it with half a map's worth of pages per query.  A goal that cannot be reached
    floods its whole component -- more tiles than that -- and the reference answers "no path" (status 1), it does not
    fail: the search ends its first pass with status 5 internally, the host sees the count when the stage's stream is
    idle and the second pass on a full-size retry slot answers.  Through the pipelined device entry point, as bench.py
    uses it, with a second batch in flight behind the first."""
    n = 4096
    e = R.Engine(n * 0.05, n * 0.05, 0.05)
    master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
    m2 = master.reshape(n, n).copy()              # [j][i]
    m2[2000:2013, 1500:1513] = 0.0
    m2[2000, 1500:1513] = m2[2012, 1500:1513] = 180.0    # a closed box, 11 x 11 free cells inside
    m2[2000:2013, 1500] = m2[2000:2013, 1512] = 180.0
    master = m2.reshape(-1).copy()
    e.upload(R.capi.LAYER_MASTER, master)
    e.astar_pipeline_depth(13)
    e.astar_configure(max_queries=256)
    q = R.synth.astar_queries(256, master, n, n, seed=2)
    inside = 2006 * n + 1506
    q["goal"][7] = inside
    hip = _Hip()
    d_q = hip.upload(q)
    bufs = [(hip.alloc(256 * 2048 * 4), hip.alloc(256 * 24)) for _ in range(2)]
    for d_paths, d_res in bufs:
        e.astar_device(d_q, 256, d_paths, 2048, d_res)
    depth, pages, mq = e.astar_effective_config()
    assert depth == 13 and mq == 256 and pages < (n // 64) * (n // 16), (depth, pages, mq)   # the premise: a share, not the whole map
    e.synchronize()
    for d_paths, d_res in bufs:
        st = hip.download(d_res, np.int32, 256 * 6).reshape(256, 6)[:, 0]
        assert st[7] == 1, st[7]                                   # "no path", not 5
        assert np.all((np.delete(st, 7) == 0) | (np.delete(st, 7) == 3)), sorted(set(st.tolist()))   # (3: path longer than the 2048 cells asked for)
    for d_paths, d_res in bufs:
        hip.free(d_paths)
        hip.free(d_res)
    hip.free(d_q)
    e.close()


class _Hip:
    """device buffers for the *_device entry points, through the HIP runtime librna.so itself links"""

    def __init__(self):
        self.h = C.CDLL("libamdhip64.so")
        self.h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.h.hipFree.argtypes = [C.c_void_p]

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.h.hipMalloc(C.byref(p), nbytes) == 0
        return p.value

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        assert self.h.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p

    def download(self, p, dtype, count):
        out = np.empty(count, dtype)
        assert self.h.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0
        return out

    def free(self, p):
        self.h.hipFree(p)


def test_astar_pipelined_batches_with_map_updates_in_between(R):
    """The bench's usage pattern: rna_update_map_device -> rna_astar_batch_device, repeated without waiting, with
    four batches in flight on rotating streams and search fields.  Every batch must see exactly the map of its own
    launch time (the neighbour-mask snapshot) and clean fields (the lazy reset of the stage it reuses)."""
    hip = _Hip()
    e = R.Engine(25.6, 25.6, 0.05)   # 512 x 512
    g = O.make_geom(25.6, 25.6, 0.05)
    ref = R.synth.obstacles_rect(e.rows, e.cols, density=0.25, seed=4)
    e.upload(R.capi.LAYER_LASER, ref)
    e.compose_master(1)
    e.astar_pipeline_depth(4)
    e.astar_configure(max_queries=48, bucket_width=6000)
    nb, nq, max_len = 7, 48, 4096
    rng = np.random.default_rng(3)
    maps, queries, outs, bufs = [], [], [], []
    for b in range(nb):
        rays = random_rays(rng, 600, 10.0, outside=0.0)
        O.himm_update(g, ref, rays)
        d_rays = hip.upload(rays)
        e.update_map_device(d_rays, len(rays), compose_mode=0)
        free = np.flatnonzero(~(np.isfinite(ref) & (ref > 0)))
        q = np.zeros(nq, R.capi.ASTAR_QUERY_DTYPE)
        q["start"], q["goal"] = rng.choice(free, nq), rng.choice(free, nq)
        d_q, d_paths, d_res = hip.upload(q), hip.alloc(nq * max_len * 4), hip.alloc(nq * 24)
        e.astar_device(d_q, nq, d_paths, max_len, d_res)          # asynchronous: no wait before the next map update
        maps.append(ref.copy()); queries.append(q); outs.append((d_paths, d_res)); bufs += [d_rays, d_q, d_paths, d_res]
    e.synchronize()
    found = 0
    for b in range(nb):
        res = hip.download(outs[b][1], np.int32, nq * 6).reshape(nq, 6)
        paths = hip.download(outs[b][0], np.int32, nq * max_len).reshape(nq, max_len)
        _, nbr = O.astar_masks(maps[b], e.rows, e.cols)
        for k in range(nq):
            ores, opath, _ = O.astar_query(nbr, e.rows, e.cols, queries[b]["start"][k], queries[b]["goal"][k])
            assert res[k, 0] == (0 if ores.status == 0 else 1), (b, k)
            if ores.status == 0:
                assert res[k, 1] == ores.path_len and res[k, 2] == ores.cost, (b, k)
                assert np.array_equal(paths[k, :ores.path_len], opath), (b, k)
                found += 1
    assert found > nb * nq // 2
    assert same_f32(e.download(R.capi.LAYER_MASTER), ref)
    for p in bufs:
        hip.free(p)
    e.close()


def test_bench_loop_at_full_size_matches_oracle_every_step(R):
    """bench.py's timed loop as it runs, at BASELINE's size: 4096 x 4096 map, per step a 100 032-ray HIMM batch (four
    batches in rotation) + fused compose -> VFH+ for 256 poses (four pose sets, the robots' VFH state carried from step
    to step) -> 256 grid-A* queries (four query sets) through the asynchronous device entry points at THE BENCH'S OWN pipeline
    depth (bench.DEFAULT_PIPELINE batches in flight on CU-masked streams: the page budget per stage, and with it the load on
    the retry path, is the headline run's), nothing waited for in between, for 3 x depth steps: every stage
    is used three times, so the lazy reset of the pages a stage's previous search handed out runs twice per stage under
    the load it has in the bench.  Every step's map, VFH+ commands and histograms and the statuses of all its A* queries,
    and the costs / paths of a rotating twelfth of them (every query of the four sets at least once over the run; all
    256 of a batch at once: test_astar_config3_every_bench_query_matches_oracle and bench.py --check-paths), against the
    oracle fed with the same sequence."""
    import bench
    hip = _Hip()
    n, nq, rot, depth, max_len = 4096, 256, 4, bench.DEFAULT_PIPELINE, 32768
    steps = -(-3 * depth // rot) * rot   # every stage three times (at least), a whole number of turns of the rotating input sets: 60 for 20 stages
    part = steps // rot          # visits of one query set; each checks every part-th query
    assert steps % rot == 0 and steps >= 3 * depth and depth >= 13
    L = n * 0.05
    e = R.Engine(L, L, 0.05)
    g = O.make_geom(L, L, 0.05)
    ref = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
    e.upload(R.capi.LAYER_LASER, ref)
    e.compose_master(1)
    ray_sets = [R.synth.rays(64, 1563, L, L, seed=4 + k) for k in range(rot)]
    pose_sets = [R.synth.poses(nq, L, L, seed=1 + k) for k in range(rot)]
    query_sets = [R.synth.astar_queries(nq, ref, n, n, seed=2 + k) for k in range(rot)]
    d_rays = [hip.upload(r) for r in ray_sets]
    d_poses = [hip.upload(p) for p in pose_sets]
    d_q = [hip.upload(q) for q in query_sets]
    e.vfh_init(nq)
    e.astar_pipeline_depth(depth)
    e.astar_configure(max_queries=nq)
    outs = []
    for s in range(steps):
        k = s % rot
        d_vfh, d_origin, d_hist = hip.alloc(nq * 16), hip.alloc(nq * 72 * 4), hip.alloc(nq * 72 * 4)
        d_paths, d_res = hip.alloc(nq * max_len * 4), hip.alloc(nq * 24)
        e.update_map_device(d_rays[k], len(ray_sets[k]), compose_mode=0)
        e.vfh_step_device(d_poses[k], nq, d_vfh, d_origin, d_hist)
        e.astar_device(d_q[k], nq, d_paths, max_len, d_res)
        outs.append((d_vfh, d_origin, d_hist, d_paths, d_res))
    e.synchronize()
    oracles = [O.OracleVfh() for _ in range(nq)]
    found = answered = checked = 0
    covered = [set() for _ in range(rot)]
    for s in range(steps):
        k = s % rot
        O.himm_update(g, ref, ray_sets[k].view(O.RAY_DTYPE))
        vout = hip.download(outs[s][0], np.uint8, nq * 16).view(R.capi.VFH_OUT_DTYPE)
        origin = hip.download(outs[s][1], np.float32, nq * 72).reshape(nq, 72)
        hist = hip.download(outs[s][2], np.float32, nq * 72).reshape(nq, 72)
        poses = pose_sets[k].copy()
        for r in range(nq):
            p = poses[r]
            cs, ct = oracles[r].step_pose(g, ref, p["x"], p["y"], p["yaw"], int(p["current_speed"]), p["goal_direction"],
                                          p["goal_distance"], p["goal_tolerance"], float(p["dt"]))
            assert (vout["chosen_speed"][r], vout["chosen_turnrate"][r]) == (cs, ct), (s, r)
            assert origin[r].tobytes() == oracles[r].origin_hist().tobytes() and hist[r].tobytes() == oracles[r].hist().tobytes(), (s, r)
        res = hip.download(outs[s][4], np.int32, nq * 6).reshape(nq, 6)
        paths = hip.download(outs[s][3], np.int32, nq * max_len).reshape(nq, max_len)
        # every query answered (found / no path) ...
        assert set(np.unique(res[:, 0]).tolist()) <= {0, 1}, (s, np.unique(res[:, 0]))
        answered += int((res[:, 0] == 0).sum())
        # ... and a rotating `part`-th of them (steps s, s + 4, ... visit the same query set, so the visits of a set
        # cover all of it) path for path against the oracle on the map as it is at this step: every query of every step would
        # be steps x 256 oracle searches of 0.1 s each -- most of the suite's time on a box with few host cores
        _, nbr = O.astar_masks(ref, n, n)
        q = query_sets[k]
        visit = s // rot
        picks = [i for i in range(nq) if i % part == visit % part]
        covered[k].update(picks)

        def one(j):
            i = picks[j]
            ores, opath, _ = O.astar_query(nbr, n, n, q["start"][i], q["goal"][i])
            assert res[i, 0] == (0 if ores.status == 0 else 1), (s, i)
            if ores.status == 0:
                assert res[i, 1] == ores.path_len and res[i, 2] == ores.cost, (s, i)
                assert np.array_equal(paths[i, :ores.path_len], opath), (s, i)
            return ores.status == 0
        found += sum(oracle_pool(one, len(picks)))
        checked += len(picks)
    assert found > checked * 0.9 and answered > steps * nq * 0.9
    assert all(len(c) == nq for c in covered)      # every query of the four sets, on the map of one of its visits
    assert tuple(e.astar_effective_config())[0] == depth
    assert same_f32(e.download(R.capi.LAYER_MASTER), ref) and same_f32(e.download(R.capi.LAYER_LASER), ref)
    for t in outs:
        for p_ in t:
            hip.free(p_)
    for p_ in d_rays + d_poses + d_q:
        hip.free(p_)
    e.close()


def test_astar_batches_larger_than_max_queries_are_chunked(R):
    """rna_astar_configure(max_queries) bounds the concurrent searches (one field each); larger batches go through
    in chunks on rotating pipeline stages, also after the configuration and the pipeline depth changed."""
    e = R.Engine(200 * 0.05, 120 * 0.05, 0.05)
    master = R.synth.obstacles_rect(e.rows, e.cols, density=0.3, seed=21)
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.astar_queries(100, master, e.rows, e.cols, seed=6)
    q["start"][7] = -1                                     # an invalid query inside a chunk
    check_astar(R, e, master, q[:96], e.ncell, settled_counts=False, max_queries=32, bucket_width=5000)
    e.astar_pipeline_depth(3)
    res, _ = check_astar(R, e, master, q[:17], e.ncell, settled_counts=False, max_queries=8)
    assert res["status"][7] == 2
    e.astar_pipeline_depth(1)
    check_astar(R, e, master, q[50:], e.ncell, max_queries=50)
    e.close()


def test_engines_release_their_hbm(R):
    """create -> plan (pipelined stages, per-stage fields, scheduler state) -> destroy, repeatedly: free HBM returns"""
    hip = _Hip()
    hip.h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.h.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    master = R.synth.obstacles_rect(256, 256, density=0.2, seed=1)
    q = R.synth.astar_queries(16, master, 256, 256, seed=1)
    rq = R.synth.rrt_queries(4, master, 256, 256, lambda i, j: (6.4 - 0.025 - 0.05 * i, 6.4 - 0.025 - 0.05 * j), seed=1, max_samples=2000)
    before = None
    for rep in range(12):
        for kernel in ("tile",):
            os.environ["RNA_ASTAR_KERNEL"] = kernel
            e = R.Engine(12.8, 12.8, 0.05)
            e.upload(R.capi.LAYER_MASTER, master)
            e.astar_configure(max_queries=16)
            e.astar(q, 4096)
            e.vfh_init(8)
            e.vfh_step(R.synth.poses(8, 12.8, 12.8, seed=2))
            e.rrt(rq)
            e.close()
        if rep == 1:
            before = free_bytes()     # after the runtime's own pools have warmed up
    os.environ.pop("RNA_ASTAR_KERNEL", None)
    assert before is not None and free_bytes() >= before - (64 << 20)


def test_astar_invalid_and_short_buffer(R):
    e = R.Engine(3.2, 3.2, 0.05)
    master = np.zeros(e.ncell, np.float32)
    e.upload(R.capi.LAYER_MASTER, master)
    q = np.zeros(3, R.capi.ASTAR_QUERY_DTYPE)
    q[0] = (-1, 5)
    q[1] = (0, e.ncell)
    q[2] = (0, e.ncell - 1)
    res, paths = e.astar(q, 16)
    assert list(res["status"]) == [2, 2, 3] and res["path_len"][2] == 64
    e.close()


# ------------------------------------------------------------------------------------------------
# the reference's own planners
# ------------------------------------------------------------------------------------------------
def test_graph_astar_matches_oracle(R):
    e = R.Engine(2.0, 2.0, 0.05)
    loc = (C.c_double * 18)()
    euv = (C.c_int * 20)()
    O.lib().og_reference_graph(loc, euv)
    V = np.array(loc).reshape(9, 2)
    E = np.array(euv, np.int32).reshape(10, 2)
    rng = np.random.default_rng(4)
    st = rng.uniform(-2, 24, (64, 4))
    st[:, 1] = rng.uniform(-2, 12, 64)
    st[:, 3] = rng.uniform(-2, 12, 64)
    plen, paths = e.graph_astar(V, E, st)
    out = (C.c_double * 64)()
    for k in range(64):
        n = O.lib().og_graph_make_plan(O.d2(st[k, 0], st[k, 1]), O.d2(st[k, 2], st[k, 3]), out, 32)
        assert plen[k] == n
        assert np.array_equal(paths[k, :n].reshape(-1), np.array(out[:2 * n]))
    e.close()


def test_rrt_matches_oracle(R):
    e = R.Engine(10.0, 10.0, 0.05)
    g = O.make_geom(10.0, 10.0, 0.05)
    master = R.synth.obstacles_rect(e.rows, e.cols, density=0.08, seed=3, side=(4, 24))
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.rrt_queries(24, master, e.rows, e.cols, e.get_position, seed=3)
    q["target"][0] = (40.0, 0.0)          # target outside the map: plan to the boundary
    q["max_samples"][1] = 3               # sample budget exhausted
    q["max_samples"][2:12] = (1, 2, 7, 8, 9, 15, 16, 17, 40, 100)   # ... at and around the speculation round sizes
    q["seed"][12:16] = (0, 2**31, 2**32 - 1, 2**31 - 1)             # srand(0) == srand(1); seeds that are negative as int32_t
    res, paths = e.rrt(q)
    assert (res["status"] == -1).sum() >= 3 and (res["status"] == 1).sum() >= 3
    for k in range(len(q)):
        check_rrt_query(res, paths, q, k, g, master)   # bit for bit in the kernel's formulation AND counts + 1e-9 m in the reference's
    e.close()


# ------------------------------------------------------------------------------------------------
# circular buffer (GridMap::move) and degenerate inputs
# ------------------------------------------------------------------------------------------------
def test_himm_and_vfh_on_a_moved_map(R):
    """After GridMap::move the buffer start index is non-zero: index<->position math wraps
    (gmc/src/GridMapMath.cpp:70-81,467-476); LineIterator walks buffer indices without unwrapping
    (reference behaviour, restated by the oracle) and getSubmap gathers across the seam."""
    e = R.Engine(12.8, 9.6, 0.05)
    g = O.make_geom(12.8, 9.6, 0.05)
    rng = np.random.default_rng(11)
    laser = R.synth.occupancy_sparse(e.rows, e.cols, seed=5, occupied=0.03)
    for l in range(3):
        e.upload(l, laser)
    ref = laser.copy()
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
    regs = (O.Region * 4)()
    mv = C.c_int(0)
    O.lib().og_move(C.byref(g), ptrs, 1, O.d2(1.37, -0.82), regs, C.byref(mv))
    assert e.move(1.37, -0.82) and tuple(e.geometry().start_index) == tuple(g.start) != (0, 0)
    rays = random_rays(rng, 1500, 3.0, outside=0.1)
    rays["sx"] += 1.37
    rays["ex"] += 1.37
    rays["sy"] -= 0.82
    rays["ey"] -= 0.82
    O.himm_update(g, ref, rays)
    e.himm_update(R.capi.LAYER_LASER, rays.view(R.capi.RAY_DTYPE))
    assert same_f32(e.download(R.capi.LAYER_LASER), ref)
    e.compose_master(1)
    poses = R.synth.poses(48, 12.8, 9.6, seed=9, margin=0.3)
    poses["x"] += 1.37
    poses["y"] -= 0.82
    e.vfh_init(len(poses))
    check_vfh_vs_oracle(R, e, g, ref, poses, steps=3)
    e.close()


def test_astar_on_a_moved_map_searches_in_map_space(R):
    """after GridMap::move buffer neighbours are not map neighbours.  The tile
    kernels search over unwrapped indices (paths cross the circular-buffer seam, never the map edge)
    and speak buffer indices at the boundary; oracle = og_astar_query_on_map."""
    e = R.Engine(12.8, 9.6, 0.05)
    g = O.make_geom(12.8, 9.6, 0.05)
    rows, cols = e.rows, e.cols
    master = R.synth.obstacles_rect(rows, cols, density=0.25, seed=8)
    for l in range(3):
        e.upload(l, master)
    ref = master.copy()
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
    regs = (O.Region * 4)()
    mv = C.c_int(0)
    for step, target in enumerate(((3.21, -1.47), (2.0, 2.6))):
        O.lib().og_move(C.byref(g), ptrs, 1, O.d2(*target), regs, C.byref(mv))
        assert e.move(*target) and tuple(e.geometry().start_index) == tuple(g.start) != (0, 0)
        # the dropped rows/columns came back as NaN (unknown = free): fill part of them with an obstacle
        rays = random_rays(np.random.default_rng(step), 400, 3.0, outside=0.0)
        rays["sx"] += target[0]; rays["ex"] += target[0]; rays["sy"] += target[1]; rays["ey"] += target[1]
        O.himm_update(g, ref, rays)
        e.update_map(rays.view(R.capi.RAY_DTYPE), compose_mode=1)
        assert same_f32(e.download(R.capi.LAYER_MASTER), ref)
        rng = np.random.default_rng(100 + step)
        free = np.flatnonzero(~(np.isfinite(ref) & (ref > 0)))
        q = np.zeros(48, R.capi.ASTAR_QUERY_DTYPE)
        q["start"] = rng.choice(free, len(q))
        q["goal"] = rng.choice(free, len(q))
        q["goal"][0] = q["start"][0]
        res, paths = e.astar(q, 4096)
        settled = e.astar_settled(len(q))
        crossed = 0
        for k in range(len(q)):
            ores, opath = O.astar_query_on_map(g, ref, q["start"][k], q["goal"][k])
            assert res["status"][k] == (0 if ores.status == 0 else 1), k
            if ores.status == 0:
                assert res["path_len"][k] == ores.path_len and res["cost"][k] == ores.cost, k
                assert np.array_equal(paths[k][:ores.path_len], opath), k
                assert settled[k] == ores.settled, k
                bi, bj = opath % rows, opath // rows
                crossed += int((np.abs(np.diff(bi)) > 1).any() or (np.abs(np.diff(bj)) > 1).any())
        assert crossed > 0      # some paths do run across the circular-buffer seam
    e.close()


# ------------------------------------------------------------------------------------------------
# message formats either side of the path
# ------------------------------------------------------------------------------------------------
def test_occupancy_grid_matches_oracle_also_on_a_moved_map(R):
    e = R.Engine(12.8, 9.6, 0.05)
    g = O.make_geom(12.8, 9.6, 0.05)
    rng = np.random.default_rng(21)
    layer = (rng.integers(0, 19, e.ncell) * 10).astype(np.float32)      # HIMM values 0..180
    layer[rng.random(e.ncell) < 0.2] = np.nan
    layer[rng.random(e.ncell) < 0.01] = -3.0
    layer[rng.random(e.ncell) < 0.01] = 300.0
    for l in range(3):
        e.upload(l, layer)
    assert np.array_equal(e.to_occupancy_grid(R.capi.LAYER_MASTER), O.to_occupancy_grid(g, layer, 0.0, 255.0))
    assert np.array_equal(e.to_occupancy_grid(R.capi.LAYER_LASER, -1.0, 100.0), O.to_occupancy_grid(g, layer, -1.0, 100.0))
    ref = layer.copy()
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
    regs = (O.Region * 4)()
    mv = C.c_int(0)
    O.lib().og_move(C.byref(g), ptrs, 1, O.d2(-2.13, 0.77), regs, C.byref(mv))
    assert e.move(-2.13, 0.77) and tuple(g.start) != (0, 0)
    want = O.to_occupancy_grid(g, ref, 0.0, 255.0)
    assert np.array_equal(e.to_occupancy_grid(R.capi.LAYER_MASTER), want)
    assert (want == -1).sum() > (np.isnan(layer) | (layer < 0)).sum()     # the dropped rows/columns are unobserved
    with pytest.raises(R.RnaError):
        e.from_occupancy_grid(R.capi.LAYER_RANGE, np.zeros(e.ncell, np.int8))   # reference resets the geometry here
    e.close()


def test_occupancy_grid_round_trip(R):
    """grid_map_ros/test/GridMapRosTest.cpp:140-184 through the device path"""
    rng = np.random.default_rng(5)
    width, height = 50, 100
    data = rng.integers(-1, 101, width * height).astype(np.int8)
    e = R.Engine(0.1 * width, 0.1 * height, 0.1, 3.0 + 0.05 * width, 6.0 + 0.05 * height)
    assert (e.rows, e.cols) == (width, height)
    e.from_occupancy_grid(R.capi.LAYER_LASER, data)
    assert same_f32(e.download(R.capi.LAYER_LASER), O.from_occupancy_grid(width, height, data))
    assert np.array_equal(e.to_occupancy_grid(R.capi.LAYER_LASER, -1.0, 100.0), data)
    e.close()


def test_hist_msg_matches_oracle(R):
    e = R.Engine(20.0, 20.0, 0.05)
    master = R.synth.occupancy_sparse(e.rows, e.cols, seed=3, occupied=0.05)
    e.upload(R.capi.LAYER_MASTER, master)
    poses = R.synth.poses(40, 20.0, 20.0, seed=4)
    e.vfh_init(len(poses))
    for _ in range(2):
        out, origin, hist = e.vfh_step(poses)
    x, y, yb, th = e.vfh_hist_msg(len(poses))
    assert origin.max() > 65535.0                                           # the uint16 narrowing is exercised
    for k in range(len(poses)):
        ox, oy, oyb, oth = O.hist_msg(hist[k], origin[k], 5)
        assert np.array_equal(x, ox) and np.array_equal(y[k], oy) and np.array_equal(yb[k], oyb) and th == oth
    e.close()


def test_scan_to_rays_matches_oracle_and_feeds_himm(R):
    """LaserScan batches -> RangeSamples on the device, every beam transformed with the sensor pose
    interpolated between tf's start and end transforms (half of the scans move while they sweep).  Ray counts, order, origins and
    ifClearEnd flags are exact; end points are float32 values built from device cos/sin (f64), so they may
    differ from the oracle's libm by one float32 ulp -- compared at 1e-6 m, and bit-exactly for the
    overwhelming majority."""
    e = R.Engine(25.6, 25.6, 0.05)
    g = O.make_geom(25.6, 25.6, 0.05)
    for inc, beams in ((None, 1081), (np.float32(0.02), 200), (np.float32(0.0005), 4000)):
        scans, ranges = R.synth.laser_scans(24, beams, 25.6, 25.6, seed=beams, angle_increment=inc)
        assert (scans["yaw_end"] != scans["yaw"]).any() and (scans["yaw_end"] == scans["yaw"]).any()   # moving and resting sensors
        n_proj = R.capi.lib().rna_scan_projected_beams(beams, float(scans["angle_increment"][0]))
        assert n_proj == (beams if scans["angle_increment"][0] >= 0.017 else len(O.simplify_scan(beams, scans["angle_increment"][0])[0]))
        want = O.scan_to_rays(scans, ranges)
        got = e.scan_to_rays(scans, ranges)
        assert len(got) == len(want) > 0
        assert np.array_equal(got["sx"], want["sx"]) and np.array_equal(got["sy"], want["sy"])
        assert np.array_equal(got["clear_end"], want["clear_end"])
        assert np.allclose(got["ex"], want["ex"], rtol=0, atol=1e-6) and np.allclose(got["ey"], want["ey"], rtol=0, atol=1e-6)
        exact = (got["ex"] == want["ex"]) & (got["ey"] == want["ey"])
        assert exact.mean() > 0.99
        if inc is None:
            assert want["clear_end"].sum() > 0                       # the index quirk is exercised
    # the rays feed the HIMM update unchanged: same map as the oracle fed with the same (device-made) rays
    ref = np.full(e.ncell, np.nan, np.float32)
    O.himm_update(g, ref, got.view(O.RAY_DTYPE))
    e.update_map(got, compose_mode=1)
    assert same_f32(e.download(R.capi.LAYER_MASTER), ref)
    with pytest.raises(R.RnaError):
        e.scan_to_rays(scans, ranges, max_rays=10)                   # RNA_ECAPACITY, not a silent truncation
    e.close()


def test_scan_to_rays_with_full_sensor_transforms_matches_oracle(R):
    """SURVEY.md 8f row 3, non-planar mounts: the same ingestion with the transforms tf reports (translation + quaternion at the
    scan's start and end) -- tilted, rolled and raised sensors, half of them moving while they sweep, end quaternions of either
    sign.  Ray counts, order, origins and ifClearEnd flags are exact; end points are float32 values built from device
    acos / sin (f64), compared at 1e-6 m and bit-exactly for the overwhelming majority.  Planar poses through this entry point
    give the planar entry point's rays within the rounding of the two formulations."""
    e = R.Engine(25.6, 25.6, 0.05)
    g = O.make_geom(25.6, 25.6, 0.05)
    for inc, beams in ((None, 1081), (np.float32(0.02), 200), (np.float32(0.0005), 4000)):
        scans, ranges = R.synth.laser_scans_tf(24, beams, 25.6, 25.6, seed=beams + 3, angle_increment=inc)
        tilted = np.abs(scans["q"][:, :2]).max(axis=1) > 1e-3
        assert tilted.any() and (~tilted).any() and (scans["q_end"] != scans["q"]).any(axis=1).any()
        want = O.scan_to_rays_tf(scans, ranges)
        got = e.scan_to_rays_tf(scans, ranges)
        assert len(got) == len(want) > 0
        assert np.array_equal(got["sx"], want["sx"]) and np.array_equal(got["sy"], want["sy"])
        assert np.array_equal(got["clear_end"], want["clear_end"])
        assert np.allclose(got["ex"], want["ex"], rtol=0, atol=1e-6) and np.allclose(got["ey"], want["ey"], rtol=0, atol=1e-6)
        exact = (got["ex"] == want["ex"]) & (got["ey"] == want["ey"])
        assert exact.mean() > 0.99
    # planar poses, both entry points
    flat, ranges = R.synth.laser_scans(16, 720, 25.6, 25.6, seed=77)
    tf = np.zeros(len(flat), R.capi.SCAN_TF_DTYPE)
    for f in ("angle_min", "angle_max", "angle_increment", "range_min", "range_max", "n_ranges", "ranges_offset"):
        tf[f] = flat[f]
    for a, b, c, d in (("x", "y", "yaw", ""), ("x_end", "y_end", "yaw_end", "_end")):
        tf["t" + d][:, 0], tf["t" + d][:, 1] = flat[a], flat[b]
        tf["q" + d][:, 2], tf["q" + d][:, 3] = np.sin(flat[c] / 2), np.cos(flat[c] / 2)
    a, b = e.scan_to_rays(flat, ranges), e.scan_to_rays_tf(tf, ranges)
    assert len(a) == len(b) > 0 and np.array_equal(a["clear_end"], b["clear_end"]) and np.array_equal(a["sx"], b["sx"])
    assert np.allclose(a["ex"], b["ex"], rtol=0, atol=2e-6) and np.allclose(a["ey"], b["ey"], rtol=0, atol=2e-6)
    # the rays feed the HIMM update unchanged
    ref = np.full(e.ncell, np.nan, np.float32)
    O.himm_update(g, ref, b.view(O.RAY_DTYPE))
    e.update_map(b, compose_mode=1)
    assert same_f32(e.download(R.capi.LAYER_MASTER), ref)
    with pytest.raises(R.RnaError):
        e.scan_to_rays_tf(tf, ranges, max_rays=10)
    e.close()


def test_scan_with_tf_jitter_between_start_and_end_keeps_every_beam(R):
    """Round 4's advisor finding: end orientation 1e-10 .. 1e-7 rad from the start one (a robot standing still under tf jitter)
    -- the slerp's acos argument rounds to 1 + ulp; clamped as tfAcos does, every beam stays finite and equals the oracle's."""
    from test_oracle_scan import _jittered_scans
    e = R.Engine(25.6, 25.6, 0.05)
    scans, still, ranges = _jittered_scans()
    got, want = e.scan_to_rays_tf(scans, ranges), O.scan_to_rays_tf(scans, ranges)
    const = e.scan_to_rays_tf(still, ranges)
    assert len(got) == len(want) == len(const) > 1000
    for f in ("sx", "sy", "ex", "ey"):
        assert np.isfinite(got[f]).all()
    assert np.array_equal(got["sx"], want["sx"]) and np.array_equal(got["clear_end"], want["clear_end"])
    assert np.allclose(got["ex"], want["ex"], rtol=0, atol=1e-6) and np.allclose(got["ey"], want["ey"], rtol=0, atol=1e-6)
    assert np.allclose(got["ex"], const["ex"], rtol=0, atol=1e-5) and np.allclose(got["ey"], const["ey"], rtol=0, atol=1e-5)
    e.close()


def test_positions_on_the_far_edge_are_outside_not_out_of_bounds(R):
    """A position within rounding of the map's far edge can divide to index == size (see
    test_oracle_gridmap.test_index_from_position_never_returns_an_index_past_the_map).  rna_get_index, HIMM ray end
    points there and RRT discs touching it must agree with the oracle -- and stay inside the layer."""
    from test_oracle_gridmap import far_edge_positions
    probes = far_edge_positions(n_geoms=60, seed=10)
    rejected = 0
    k = 0
    while k < len(probes):
        g = probes[k][0]
        same = [p for gg, p in probes[k:k + 12] if gg is g]
        k += len(same)
        e = R.Engine(g.len[0], g.len[1], g.res, g.pos[0], g.pos[1])
        assert (e.rows, e.cols) == (g.size[0], g.size[1])
        rays = np.zeros(len(same), O.RAY_DTYPE)
        for i, p in enumerate(same):
            idx = O.i2(0, 0)
            ok = O.lib().og_index_from_position(C.byref(g), O.d2(*p), idx)
            got = e.get_index(*p)
            assert (got is not None) == bool(ok) and (not ok or got == (idx[0], idx[1])), p
            rejected += (not ok) and bool(O.lib().og_position_within_map(O.d2(*p), g.len, g.pos))
            rays[i] = (g.pos[0], g.pos[1], p[0], p[1], 0, 0)       # a hit exactly there
        init = np.zeros(e.ncell, np.float32)
        e.upload(R.capi.LAYER_LASER, init)
        ref = init.copy()
        O.himm_update(g, ref, rays)
        e.himm_update(R.capi.LAYER_LASER, rays.view(R.capi.RAY_DTYPE))
        assert same_f32(e.download(R.capi.LAYER_LASER), ref)
        e.close()
    assert rejected > 0


def test_empty_batches_are_noops(R):
    e = R.Engine(3.2, 3.2, 0.05)
    e.himm_update(R.capi.LAYER_LASER, np.zeros(0, R.capi.RAY_DTYPE))
    e.update_map(np.zeros(0, R.capi.RAY_DTYPE))
    e.vfh_init(4)
    out, origin, hist = e.vfh_step(np.zeros(0, R.capi.POSE_DTYPE))
    assert len(out) == 0
    res, paths = e.astar(np.zeros(0, R.capi.ASTAR_QUERY_DTYPE), 8)
    assert len(res) == 0
    res, paths = e.rrt(np.zeros(0, R.capi.RRT_QUERY_DTYPE))
    assert len(res) == 0
    assert np.all(np.isnan(e.download(R.capi.LAYER_MASTER)))
    with pytest.raises(R.RnaError):
        e.vfh_step(np.zeros(5, R.capi.POSE_DTYPE))     # more poses than VFH instances
    e.close()


def test_astar_large_grid_properties(R):
    """BASELINE.json's full size (4096 x 4096): properties that need no oracle run -- every path is a
    chain of legal 8-connected moves over free cells whose 1000/1414 cost equals the reported cost,
    the reverse query has the same cost, and two bucket widths give identical paths."""
    n = 4096
    e = R.Engine(n * 0.05, n * 0.05, 0.05)
    master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.astar_queries(24, master, n, n, seed=5)
    rq = q.copy()
    rq["start"], rq["goal"] = q["goal"], q["start"]
    e.astar_configure(max_queries=24, bucket_width=8000)
    res, paths = e.astar(q, 32768)
    rres, _ = e.astar(rq, 32768)
    e.astar_configure(bucket_width=3000)
    res2, paths2 = e.astar(q, 32768)
    nbr = e.nbr_mask()
    assert np.all(res["status"] == 0) and np.array_equal(res["cost"], rres["cost"])
    assert np.array_equal(res["cost"], res2["cost"]) and np.array_equal(res["path_len"], res2["path_len"])
    off = np.array([-1 - n, -n, 1 - n, -1, 1, n - 1, n, n + 1])
    cost = np.array([1414, 1000, 1414, 1000, 1000, 1414, 1000, 1414])
    for k in range(len(q)):
        p = paths[k, :res["path_len"][k]].astype(np.int64)
        assert np.array_equal(p, paths2[k, :res2["path_len"][k]])
        assert p[0] == q["start"][k] and p[-1] == q["goal"][k]
        d = p[1:] - p[:-1]
        kk = np.searchsorted(off, d)
        assert np.all(off[np.clip(kk, 0, 7)] == d)                      # only the 8 neighbour offsets
        assert np.all((nbr[p[:-1]] >> kk) & 1)                          # each move allowed by the mask
        assert int(cost[kk].sum()) == res["cost"][k]
    # and the first few queries against the CPU oracle at this size: path, cost, settled count
    settled = e.astar_settled(len(q))
    _, onbr = O.astar_masks(master, n, n)
    assert np.array_equal(onbr, nbr)
    gw = np.empty(n * n, np.int32)
    for k in range(6):
        ores, opath, _ = O.astar_query(onbr, n, n, q["start"][k], q["goal"][k], g_work=gw)
        assert ores.status == 0 and ores.cost == res2["cost"][k] and ores.settled == settled[k]
        assert np.array_equal(opath, paths2[k, :res2["path_len"][k]])
    e.close()


# ------------------------------------------------------------------------------------------------
# BASELINE.json's full sizes for the rows whose oracle is fast enough to run them whole
# ------------------------------------------------------------------------------------------------
def test_astar_config3_every_bench_query_matches_oracle(R):
    """Config 3 as bench.py serves it: the 4096 x 4096 rectangle map (seed 2) and ALL 256 (start, goal) pairs of a
    batch, through the asynchronous device entry point with batches in flight: path, cost and settled count E of
    every query against the CPU oracle."""
    n = 4096
    e = R.Engine(n * 0.05, n * 0.05, 0.05)
    master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.astar_queries(256, master, n, n, seed=2)
    e.astar_configure(max_queries=256)
    res, paths = e.astar(q, 32768)
    settled = e.astar_settled(len(q))
    _, onbr = O.astar_masks(master, n, n)
    assert np.array_equal(onbr, e.nbr_mask())

    def one(k):
        ores, opath, _ = O.astar_query(onbr, n, n, q["start"][k], q["goal"][k])
        assert ores.status == res["status"][k] == 0, k
        assert ores.cost == res["cost"][k] and ores.settled == settled[k] and ores.path_len == res["path_len"][k], k
        assert np.array_equal(opath, paths[k, :ores.path_len]), k
        return ores.settled
    total = sum(oracle_pool(one, len(q)))
    assert total > 5e7
    e.close()


def test_rrt_config4_one_gpu_share_matches_oracle(R):
    """Config 4: 2048 x 2048, 30 % rectangles (seed 3), one GPU's share of the 4096 trees = 512 queries (the reference's
    2000 extendTree iterations, rrt_planner.cpp:6, each bounded to the batch's 100 000 samples in total): status, tree
    size, sample count, path length and every way point bit for bit against the oracle with the steering step in the kernel's
    formulation (near + 0.4 (dx, dy) / sqrt(dx^2 + dy^2); oracle/rrt.c `steer` = 1) AND, for every one of the 512 queries, the same
    counts and way points within 1e-9 m against the reference's own atan2 / cos / sin formulation (`steer` = 0,
    rrt_planner.cpp:43-51): check_rrt_query.  tests/test_oracle_misc.py runs the same 512 queries CPU against CPU."""
    n = 2048
    L = n * 0.05
    e = R.Engine(L, L, 0.05)
    g = O.make_geom(L, L, 0.05)
    master = R.synth.obstacles_rect(n, n, density=0.30, seed=3)
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.rrt_queries(512, master, n, n, e.get_position, seed=3, max_samples=100000)
    res, paths = e.rrt(q)

    st = oracle_pool(lambda k: check_rrt_query(res, paths, q, k, g, master), len(q))
    assert 0 in st                       # some trees reach their target within the budget
    e.close()



def test_himm_full_ray_batch_on_4096_matches_oracle(R):
    """Config 5's ray batch (100 032 rays from 64 origins, 1-6 m, 80 % hits) on the 4096 x 4096 bench map, applied
    three times in a row (steady state: marks saturate, clears and marks interleave on shared cells): bit-exact."""
    n = 4096
    L = n * 0.05
    e = R.Engine(L, L, 0.05)
    g = O.make_geom(L, L, 0.05)
    laser = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
    e.upload(R.capi.LAYER_LASER, laser)
    ref = laser.copy()
    for rep in range(3):
        rays = R.synth.rays(64, 1563, L, L, seed=4 + rep)
        O.himm_update(g, ref, rays.view(O.RAY_DTYPE))
        e.update_map(rays, compose_mode=0)
    assert same_f32(e.download(R.capi.LAYER_LASER), ref)
    assert same_f32(e.download(R.capi.LAYER_MASTER), ref)       # fused dirty-tile compose == whole-layer copy
    e.close()


def test_himm_pair_buffer_below_the_worst_case_is_checked_and_grows(R, monkeypatch):
    """The (tile, ray) pair buffer of the rasteriser is allocated for the worst case (a ray registered with two tiles per
    64-cell tile column of the map) only while that is small; beyond a limit it starts at 16 pairs per ray, every batch's
    exact pair count is read back before the pairs are written, and the buffer grows when a batch needs more.  Forced
    here with a tiny limit: short rays first (fit), then rays across the whole map (20 tile columns each: must grow)."""
    monkeypatch.setenv("RNA_HIMM_PAIRS_LIMIT", "4096")
    n = 1280
    L = n * 0.05
    e = R.Engine(L, L, 0.05)
    g = O.make_geom(L, L, 0.05)
    laser = R.synth.obstacles_rect(n, n, density=0.2, seed=5)
    e.upload(R.capi.LAYER_LASER, laser)
    ref = laser.copy()
    short = R.synth.rays(4, 250, L, L, seed=11, lmax=3.0)
    rng = np.random.default_rng(12)
    long_ = np.zeros(1000, R.capi.RAY_DTYPE)                      # border to border, every direction
    a = rng.uniform(0, 2 * np.pi, len(long_))
    long_["sx"], long_["sy"] = 0.49 * L * np.cos(a), 0.49 * L * np.sin(a)
    long_["ex"], long_["ey"] = -long_["sx"] + rng.uniform(-1, 1, len(long_)), -long_["sy"] + rng.uniform(-1, 1, len(long_))
    long_["clear_end"] = rng.integers(0, 2, len(long_))
    for rays in (short, long_, short, long_):
        O.himm_update(g, ref, rays.view(O.RAY_DTYPE))
        e.update_map(rays, compose_mode=0)
        assert same_f32(e.download(R.capi.LAYER_LASER), ref)
    e.close()


def test_vfh_config2_full_batch_matches_oracle(R):
    """Config 2 as BASELINE.json states it: 1024 x 1024 grid (2 % occupied, 10 % unknown), 1024 poses, Steerer
    parameters, two consecutive steps (the second one exercises the stateful binary histogram)."""
    n = 1024
    L = n * 0.05
    e = R.Engine(L, L, 0.05)
    g = O.make_geom(L, L, 0.05)
    master = R.synth.occupancy_sparse(n, n, seed=1)
    e.upload(R.capi.LAYER_MASTER, master)
    poses = R.synth.poses(1024, L, L, seed=1)
    e.vfh_init(len(poses))
    check_vfh_vs_oracle(R, e, g, master, poses, steps=2)
    e.close()


def test_rrt_config4_sample_matches_oracle(R):
    """Config 4's map (2048 x 2048, 30 % rectangles, seed 3): a sample of its queries against the oracle --
    status, tree size, sample count and way points exact with steer = 1, counts + 1e-9 m with steer = 0 (check_rrt_query)."""
    n = 2048
    L = n * 0.05
    e = R.Engine(L, L, 0.05)
    g = O.make_geom(L, L, 0.05)
    master = R.synth.obstacles_rect(n, n, density=0.30, seed=3)
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.rrt_queries(24, master, n, n, e.get_position, seed=3, max_samples=20000)
    res, paths = e.rrt(q)
    for k in range(len(q)):
        check_rrt_query(res, paths, q, k, g, master)
    e.close()


# ------------------------------------------------------------------------------------------------
# a7 as an entry point of its own: MapProvider::getSubMap / GridMap::getSubmap
# ------------------------------------------------------------------------------------------------
def test_get_submap_matches_oracle_also_on_a_moved_map(R):
    rng = np.random.default_rng(17)
    e = R.Engine(12.0, 9.0, 0.05, 1.0, -2.0)           # 240 x 180
    g = O.make_geom(12.0, 9.0, 0.05, 1.0, -2.0)
    ref = rng.choice(np.array([np.nan, 0, 30, 60, 180, -1, 7.25], np.float32), e.ncell)
    for l in range(3):
        e.upload(l, ref)
    sub_g = O.Geom()
    buf = np.empty(e.ncell, np.float32)
    checked = failed = 0
    for moved in range(3):
        if moved:
            target = (1.0 + rng.uniform(-3, 3), -2.0 + rng.uniform(-3, 3))
            ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
            regs = (O.Region * 4)()
            mv = C.c_int(0)
            O.lib().og_move(C.byref(g), ptrs, 1, O.d2(*target), regs, C.byref(mv))
            e.move(*target)
            assert tuple(e.geometry().start_index) == tuple(g.start) and g.start[0] + g.start[1] > 0
        cx, cy = float(g.pos[0]), float(g.pos[1])
        cases = [(cx, cy, 1.5, 1.5), (cx, cy, 0.0, 0.0), (cx, cy, 1e6, 1e6), (cx + 5.9, cy - 4.4, 1.5, 1.5),
                 (cx - 6.0, cy + 4.5, 2.0, 2.0), (cx + 40.0, cy, 1.5, 1.5), (cx, cy, 12.0, 9.0), (cx, cy, 0.05, 0.05)]
        for _ in range(60):
            cases.append((cx + rng.uniform(-7, 7), cy + rng.uniform(-5.5, 5.5), rng.uniform(0, 11), rng.uniform(0, 11)))
        for (px, py, lx, ly) in cases:
            ok = O.lib().og_get_submap(C.byref(g), O.fptr(ref), O.d2(px, py), O.d2(lx, ly), C.byref(sub_g), O.fptr(buf),
                                       buf.size)
            got = e.get_submap(R.capi.LAYER_LASER, px, py, lx, ly)
            assert (got is not None) == bool(ok), (moved, px, py, lx, ly)
            if not ok:
                failed += 1
                continue
            info, data = got
            inf = O.SubmapInfo()
            assert O.lib().og_submap_information(C.byref(g), O.d2(px, py), O.d2(lx, ly), C.byref(inf))
            assert tuple(info.size) == tuple(sub_g.size) and tuple(info.top_left) == tuple(inf.top_left)
            assert tuple(info.position) == tuple(sub_g.pos) and tuple(info.length) == tuple(sub_g.len)
            assert same_f32(data, buf[:sub_g.size[0] * sub_g.size[1]]), (moved, px, py, lx, ly)
            checked += 1
    assert checked > 120 and failed > 10     # requested centres outside the clamped window fail, as in the reference
    # the device variant writes the same cells; a buffer that is too small is an error, not a truncation
    import torch
    info = R.capi.SubmapInfo()
    t = torch.zeros(31 * 31, dtype=torch.float32, device="cuda")
    cx, cy = float(g.pos[0]), float(g.pos[1])
    rc = R.capi.lib().rna_get_submap_device(e.h, R.capi.LAYER_LASER, cx, cy, 1.5, 1.5, t.data_ptr(), t.numel(), C.byref(info))
    e.synchronize()
    assert rc == 1
    _, host = e.get_submap(R.capi.LAYER_LASER, cx, cy, 1.5, 1.5)
    assert same_f32(t.cpu().numpy()[:len(host)], host)
    rc = R.capi.lib().rna_get_submap_device(e.h, R.capi.LAYER_LASER, cx, cy, 3.0, 3.0, t.data_ptr(), t.numel(), C.byref(info))
    assert rc == -4 and info.size[0] * info.size[1] > t.numel()      # RNA_ECAPACITY, info still filled
    e.close()


# ------------------------------------------------------------------------------------------------
# A.6 plan following feeding the VFH+ step: Steerer::acceptPlan + Steerer::update for a fleet
# ------------------------------------------------------------------------------------------------
def test_steerer_loop_follow_plan_then_vfh_matches_oracle(R):
    rng = np.random.default_rng(23)
    e = R.Engine(20.0, 20.0, 0.05)
    g = O.make_geom(20.0, 20.0, 0.05)
    master = R.synth.occupancy_sparse(e.rows, e.cols, seed=8, occupied=0.01)
    e.upload(R.capi.LAYER_MASTER, master)
    n = 24
    e.vfh_init(n)
    oracles = [O.OracleVfh() for _ in range(n)]
    plans = [np.cumsum(rng.normal(scale=0.35, size=(int(rng.integers(2, 14)), 2)), 0) + rng.uniform(-6, 6, 2) for _ in range(n)]
    pos = np.array([p[0] for p in plans]) + rng.normal(scale=0.05, size=(n, 2))
    yaw = rng.uniform(-np.pi, np.pi, n)
    idx_c, idx_o = [1] * n, [1] * n                    # acceptPlan: planIndex_ = 1
    speed = np.zeros(n)
    active = list(range(n))
    steps = 0
    while active and steps < 40:
        poses = np.zeros(len(active), R.capi.POSE_DTYPE)
        still = []
        for k in active:
            ok, idx_c[k], pose = R.capi.follow_plan(plans[k], idx_c[k], pos[k, 0], pos[k, 1], yaw[k], speed[k] / 1000.0, 0.2)
            ook, idx_o[k], ang, dist = O.follow_plan(plans[k], idx_o[k], pos[k, 0], pos[k, 1], yaw[k])
            assert (ok, idx_c[k]) == (ook, idx_o[k])
            if ok:
                assert (np.float32(pose["goal_direction"]), np.float32(pose["goal_distance"])) == (ang, dist)
                poses[len(still)] = pose
                still.append(k)
        active = still
        if not active:
            break
        poses = poses[:len(active)]
        # the engine's VFH instances are addressed by batch position: keep robot k on instance k
        full = np.zeros(n, R.capi.POSE_DTYPE)
        full["dt"] = 0.2
        full["goal_distance"] = 3000.0
        full["goal_tolerance"] = 250.0
        full["x"], full["y"], full["yaw"] = pos[:, 0], pos[:, 1], yaw
        full["goal_direction"] = 90.0
        full[active] = poses
        out, origin, hist = e.vfh_step(full)
        for k in range(n):
            p = full[k]
            cs, ct = oracles[k].step_pose(g, master, p["x"], p["y"], p["yaw"], int(p["current_speed"]), p["goal_direction"],
                                          p["goal_distance"], p["goal_tolerance"], float(p["dt"]))
            assert (out["chosen_speed"][k], out["chosen_turnrate"][k]) == (cs, ct), (steps, k)
            assert bits(origin[k]).tobytes() == bits(oracles[k].origin_hist()).tobytes(), (steps, k)
        # Steerer::pubVel: the robots drive with the chosen command for one period
        speed = out["chosen_speed"].astype(np.float64)
        yaw = yaw + np.radians(out["chosen_turnrate"]) * 0.2
        pos = pos + (speed / 1000.0 * 0.2)[:, None] * np.stack([np.cos(yaw), np.sin(yaw)], 1)
        for k in active:                                 # and are nudged towards their way point so that plans finish
            pos[k] += 0.5 * (plans[k][idx_c[k]] - pos[k])
        steps += 1
    assert steps >= 3 and len(active) < n
    e.close()


def test_submap_engine_is_getsubmap_as_a_gridmap_and_plans_like_nav_makeplan(R):
    """Nav::makePlan (mc/src/nav_node.cpp:136-152): getSubMap -> a GridMap of the planning window -> RrtPlanner on it."""
    rng = np.random.default_rng(29)
    e = R.Engine(16.0, 12.0, 0.05, -1.0, 2.0)
    g = O.make_geom(16.0, 12.0, 0.05, -1.0, 2.0)
    layers = [R.synth.obstacles_rect(e.rows, e.cols, density=0.10, seed=31 + l, side=(2, 10)) for l in range(3)]
    for l in range(3):
        e.upload(l, layers[l])
    ptrs = (C.POINTER(C.c_float) * 3)(*[O.fptr(a) for a in layers])
    regs = (O.Region * 4)()
    mv = C.c_int(0)
    O.lib().og_move(C.byref(g), ptrs, 3, O.d2(1.35, 0.2), regs, C.byref(mv))    # planning on a recentred map
    assert e.move(1.35, 0.2) and mv.value
    cx, cy = float(g.pos[0]), float(g.pos[1])
    for (px, py, lx, ly) in ((cx, cy, 10.0, 10.0), (cx + 6.0, cy - 4.0, 6.0, 6.0), (cx - 7.9, cy + 5.9, 4.0, 3.0)):
        sub = e.submap_engine(px, py, lx, ly)
        sg = O.Geom()
        want = [np.empty(e.ncell, np.float32) for _ in range(3)]
        for l in range(3):
            assert O.lib().og_get_submap(C.byref(g), O.fptr(layers[l]), O.d2(px, py), O.d2(lx, ly), C.byref(sg), O.fptr(want[l]), e.ncell)
        gg = sub.geometry()
        assert tuple(gg.size) == tuple(sg.size) and tuple(gg.position) == tuple(sg.pos) and tuple(gg.length) == tuple(sg.len)
        assert tuple(gg.start_index) == (0, 0) and gg.resolution == g.res
        n = sg.size[0] * sg.size[1]
        for l in range(3):
            assert same_f32(sub.download(l), want[l][:n]), (px, py, l)
        # RrtPlanner(mapForPlan, start, target): targets inside and outside the window (finish at its border)
        q = np.zeros(6, R.capi.RRT_QUERY_DTYPE)
        free = np.flatnonzero(~(np.nan_to_num(want[0][:n], nan=0.0) > 0))
        for k in range(len(q)):
            c = int(rng.choice(free))
            q["start"][k] = sub.get_position(c % sg.size[0], c // sg.size[0])
            q["target"][k] = (px + rng.uniform(-0.8, 0.8) * lx, py + rng.uniform(-0.8, 0.8) * ly)
        q["close_tolerance"] = 0.2
        q["seed"] = np.arange(1, len(q) + 1)
        q["max_samples"] = 20000
        res, paths = sub.rrt(q)
        for k in range(len(q)):
            ores, opath = O.rrt_plan(sg, want[0][:n].copy(), q["start"][k], q["target"][k], 0.2, int(q["seed"][k]), 20000)
            assert (res["status"][k], res["tree_size"][k], res["samples"][k], res["path_len"][k]) == \
                   (ores.status, ores.tree_size, ores.samples, ores.path_len), (px, py, k)
            assert np.allclose(paths[k, :ores.path_len], opath, rtol=0, atol=1e-9)
        sub.close()
    assert e.submap_engine(cx + 300.0, cy, 2.0, 2.0) is None      # isSuccess == false
    e.close()
