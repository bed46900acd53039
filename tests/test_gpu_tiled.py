"""Tiled single-map mode on the GPU (SURVEY.md 8e mode 2, BASELINE config 5): the window-restricted
HIMM update against the oracle's whole-map update, and the complete tiled loop (windowed HIMM + compose
-> halo exchange -> VFH+ for the owned poses -> all-gather -> sharded grid A*) run by two ranks that
share the one GPU of the test box (gloo carries the messages there; RCCL needs one device per rank)."""
import os
import socket
import sys

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same_f32(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(bits(a)[~np.isnan(a)], bits(b)[~np.isnan(b)])


def test_windowed_himm_is_the_whole_map_update_inside_the_window():
    import ros_navigation_amd as R
    from ros_navigation_amd import dist as D
    rows, cols = 300, 260
    e = R.Engine(rows * 0.05, cols * 0.05, 0.05)
    g = O.make_geom(rows * 0.05, cols * 0.05, 0.05)
    rng = np.random.default_rng(11)
    before = rng.choice(np.array([np.nan, 0, 10, 50, 150, 160, 170, 180, 7.5, -3], np.float32), e.ncell)
    rays = R.synth.rays(12, 900, rows * 0.05, cols * 0.05, seed=12, lmin=0.2, lmax=7.0, margin=0.5)
    # many marks on a few cells that sit on both sides of a window border (row 150 = first row of tile a=1)
    for k, i in enumerate((148, 149, 150, 151)):
        x, y = e.get_position(i, 65 * 2)
        rays["ex"][k * 40:(k + 1) * 40] = x
        rays["ey"][k * 40:(k + 1) * 40] = y
        rays["clear_end"][k * 40:(k + 1) * 40] = 0
    after = before.copy()
    O.himm_update(g, after, rays.view(O.RAY_DTYPE))
    L = D.TileLayout.for_world(rows, cols, 8)
    B, A = before.reshape(cols, rows), after.reshape(cols, rows)
    for rank in range(8):
        i0, ni, j0, nj = L.window(rank)
        e.upload(R.capi.LAYER_LASER, before)
        e.himm_set_window(i0, j0, ni, nj)
        e.himm_update(R.capi.LAYER_LASER, rays)
        got = e.download(R.capi.LAYER_LASER).reshape(cols, rows)
        want = B.copy()
        want[j0:j0 + nj, i0:i0 + ni] = A[j0:j0 + nj, i0:i0 + ni]
        assert same_f32(got, want), rank
        # region pack / unpack round trip of exactly that window
        t = e.pack_region(R.capi.LAYER_LASER, i0, ni, j0, nj)
        assert same_f32(t.cpu().numpy().reshape(nj, ni), A[j0:j0 + nj, i0:i0 + ni])
        e.fill(R.capi.LAYER_RANGE, 0.0)
        e.unpack_region(R.capi.LAYER_RANGE, i0, ni, j0, nj, t)
        z = np.zeros((cols, rows), np.float32)
        z[j0:j0 + nj, i0:i0 + ni] = A[j0:j0 + nj, i0:i0 + ni]
        assert same_f32(e.download(R.capi.LAYER_RANGE).reshape(cols, rows), z)
    e.himm_set_window()                      # whole map again
    e.upload(R.capi.LAYER_LASER, before)
    e.himm_update(R.capi.LAYER_LASER, rays)
    assert same_f32(e.download(R.capi.LAYER_LASER), after)
    with pytest.raises(R.capi.RnaError):
        e.himm_set_window(0, 0, rows + 1, 4)
    e.close()


SMALL = dict(rows=256, cols=192, rounds=3, ray_poses=10, rays_per_pose=700, lmin=0.3, margin=0.4, density=0.12, side=(2, 10),
             poses=64, edge_poses=32, queries=24, max_path=4096)
# BASELINE config 5 at full size, two of its ranks: 8192 x 8192, the 100 032-ray batch, 64 VFH+ poses (half of them on
# the window border, so their 1.5 m submaps reach into the halo), 16 A* queries after the dirty-tile gather
CONFIG5 = dict(rows=8192, cols=8192, rounds=1, ray_poses=64, rays_per_pose=1563, lmin=1.0, margin=6.5, density=0.30, side=(4, 64),
               poses=32, edge_poses=32, queries=16, max_path=32768)


def _tiled_worker(rank, world, port, out, mode="full", cfg=None):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        import torch
        import _oracle as Or
        import ros_navigation_amd as R
        from ros_navigation_amd import dist as D
        torch.cuda.set_device(0)
        dist = D.init("gloo")
        cfg = cfg or SMALL
        results = {}
        for mode in (mode if isinstance(mode, tuple) else (mode,)):   # several modes in one pair of processes: a spawn costs a torch import each
            rows, cols, res = cfg["rows"], cfg["cols"], 0.05
            lx, ly = rows * res, cols * res
            e = R.Engine(lx, ly, res)
            e.astar_pipeline_depth(1)
            e.astar_configure(max_queries=8)
            g = Or.make_geom(lx, ly, res)
            L = D.TileLayout.for_world(rows, cols, world)
            halo = D.vfh_halo(res)
            i0, ni, j0, nj = L.window(rank)
            full = R.synth.obstacles_rect(rows, cols, density=cfg["density"], seed=21, side=cfg["side"])
            e.upload(R.capi.LAYER_LASER, full)
            e.compose_master(1)
            e.himm_set_window(i0, j0, ni, nj)
            # poses: uniform ones plus a row of robots right on both sides of every window border
            poses = R.synth.poses(cfg["poses"], lx, ly, seed=3, margin=0.9)
            edge = R.synth.poses(cfg["edge_poses"], lx, ly, seed=4, margin=0.9)
            for k in range(len(edge)):                      # the same poses on every rank
                wi0, wni = L.window(k % world)[:2]
                bi = (wi0 if k % 2 else wi0 + wni - 1) + (k % 5) - 2
                bi = min(max(bi, 18), rows - 19)
                edge["x"][k] = e.get_position(bi, 0)[0]
            poses = np.concatenate([poses, edge])
            idx = np.array([e.get_index(p["x"], p["y"]) for p in poses])
            mine = poses[L.owner(idx[:, 0], idx[:, 1]) == rank].copy()
            e.vfh_init(len(mine))
            oracles = [Or.OracleVfh() for _ in range(len(mine))]
            checked = {"vfh": 0, "astar": 0, "halo_bytes": 0, "gather_bytes": 0}
            for rnd in range(cfg["rounds"]):
                rays = R.synth.rays(cfg["ray_poses"], cfg["rays_per_pose"], lx, ly, seed=30 + rnd, lmin=cfg["lmin"], lmax=6.0,
                                    margin=cfg["margin"])
                Or.himm_update(g, full, rays.view(Or.RAY_DTYPE))          # whole-map truth (laser == master)
                e.update_map(rays, compose_mode=0)                         # this rank's window only
                checked["halo_bytes"] += D.exchange_halo(e, R.capi.LAYER_MASTER, L, rank, halo, dist, tracked=(mode == "dirty"))
                vout, origin, hist = e.vfh_step(mine)
                for k in range(len(mine)):
                    p = mine[k]
                    cs, ct = oracles[k].step_pose(g, full, p["x"], p["y"], p["yaw"], int(p["current_speed"]),
                                                  p["goal_direction"], p["goal_distance"], p["goal_tolerance"], float(p["dt"]))
                    assert (vout["chosen_speed"][k], vout["chosen_turnrate"][k]) == (cs, ct), (rnd, k)
                    assert origin[k].tobytes() == oracles[k].origin_hist().tobytes(), (rnd, k)
                    assert hist[k].tobytes() == oracles[k].hist().tobytes(), (rnd, k)
                    checked["vfh"] += 1
                mine["current_speed"] = vout["chosen_speed"]
                if mode == "dirty":      # only the tiles this update changed travel; masks of exactly those are refreshed
                    checked["gather_bytes"] += D.gather_dirty(e, (R.capi.LAYER_LASER, R.capi.LAYER_MASTER), L, rank, dist)
                    e.compose_master(0)
                    lz = e.download(R.capi.LAYER_LASER)
                    assert np.array_equal(np.isnan(lz), np.isnan(full)) and np.array_equal(lz[~np.isnan(lz)], full[~np.isnan(full)]), rnd
                else:
                    checked["gather_bytes"] += D.gather_layer(e, R.capi.LAYER_MASTER, L, rank, dist)
                got = e.download(R.capi.LAYER_MASTER)
                assert np.array_equal(np.isnan(got), np.isnan(full)) and np.array_equal(got[~np.isnan(got)], full[~np.isnan(full)]), rnd
                queries = R.synth.astar_queries(cfg["queries"], full, rows, cols, seed=40 + rnd)
                lo, hi = D.shard_bounds(len(queries), rank, world)
                res_, paths = e.astar(queries[lo:hi], cfg["max_path"])
                _, nbr = Or.astar_masks(full, rows, cols)
                assert np.array_equal(e.nbr_mask(), nbr), rnd
                gw = np.empty(rows * cols, np.int32)
                for k, q in enumerate(queries[lo:hi]):
                    ores, opath, _ = Or.astar_query(nbr, rows, cols, q["start"], q["goal"], g_work=gw)
                    assert res_["status"][k] == ores.status, (rnd, k)
                    if ores.status == 0:
                        assert res_["cost"][k] == ores.cost and np.array_equal(paths[k, :ores.path_len], opath), (rnd, k)
                    checked["astar"] += 1
            e.close()
            results[mode] = checked
        dist.barrier()
        dist.destroy_process_group()
        out.put((rank, "ok", results, len(mine)))
    except BaseException as ex:  # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        out.put((rank, "fail", traceback.format_exc(), repr(ex)))


def test_tiled_loop_two_ranks_on_one_gpu_matches_the_whole_map_oracle():
    """both hand-over modes (whole owner windows / only the changed tiles), one after the other in ONE pair of processes"""
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_tiled_worker, args=(r, world, port, out, ("full", "dirty"))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[2]
    assert sum(r[3] for r in res) == 96                       # every pose served by exactly one rank
    for mode in ("full", "dirty"):
        assert all(r[2][mode]["vfh"] == 3 * r[3] and r[2][mode]["astar"] == 36 for r in res)
        # 2 x 1 layout: one 16-row strip of 192 columns per rank per round, the other window per gather
        assert all(r[2][mode]["halo_bytes"] == 3 * 16 * 192 * 4 for r in res)
        if mode == "full":
            assert all(r[2][mode]["gather_bytes"] == 3 * 128 * 192 * 4 for r in res)
        else:                                                     # whole 64 x 64 tiles, and fewer bytes than the windows
            assert all(r[2][mode]["gather_bytes"] % (4096 * 4) == 0 and 0 < r[2][mode]["gather_bytes"] <= 3 * 128 * 192 * 4 for r in res)


def test_config5_full_size_two_ranks_vfh_and_astar_legs():
    """BASELINE config 5's VFH+ and A* legs at the full 8192 x 8192 size: two of the ranks (2 x 1 windows of 4096 x 8192,
    sharing the test box's GPU) run windowed HIMM of the 100 032-ray batch + compose -> 16-cell halo exchange -> VFH+ for
    their 64 poses (half of them on the window border) -> dirty-tile gather -> 16 sharded A* queries; every output
    against the whole-map oracle, and laser / master / neighbour masks equal on both ranks after the gather."""
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_tiled_worker, args=(r, world, port, out, "dirty", CONFIG5)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[2]
    assert sum(r[3] for r in res) == 64 and all(r[3] >= 16 for r in res)
    assert all(r[2]["dirty"]["vfh"] == r[3] and r[2]["dirty"]["astar"] == 8 for r in res)
    assert all(r[2]["dirty"]["halo_bytes"] == 16 * 8192 * 4 for r in res)          # one 16-row strip of 8192 columns
    assert all(0 < r[2]["dirty"]["gather_bytes"] < 4096 * 8192 * 4 // 8 for r in res)   # dirty tiles only: far less than a window


def test_config5_full_size_windowed_himm_union_is_the_whole_map_update():
    """BASELINE config 5 at full size: 8192 x 8192 map tiled 2 x 4 (tile 4096 rows x 2048 cols), the 100 032-ray batch
    (64 origins x 1563 rays).  Every GPU's windowed update, put together, is the oracle's whole-map update."""
    import ros_navigation_amd as R
    from ros_navigation_amd import dist as D
    n = 8192
    length = n * 0.05
    e = R.Engine(length, length, 0.05)
    g = O.make_geom(length, length, 0.05)
    L = D.TileLayout.for_world(n, n, 8)
    assert L.window(0) == (0, 4096, 0, 2048)
    before = np.zeros(e.ncell, np.float32)
    before[::7] = 60.0                          # something for the clears to act on
    before[::11] = np.nan
    rays = R.synth.rays(64, 1563, length, length, seed=4)
    # origins drawn over the whole map put most rays inside one window: add rays that cross the window borders
    cross = R.synth.rays(64, 40, length, length, seed=5, lmin=1.0, lmax=6.0)
    for k in range(64):
        i0, ni, j0, nj = L.window(k % 8)
        x, y = e.get_position(i0 + (ni - 1 if k & 8 else 0), j0 + (nj - 1 if k & 16 else 0))
        sl = slice(k * 40, (k + 1) * 40)
        dx, dy = cross["ex"][sl] - cross["sx"][sl], cross["ey"][sl] - cross["sy"][sl]
        cross["sx"][sl], cross["sy"][sl] = x - 0.5 * dx, y - 0.5 * dy
        cross["ex"][sl], cross["ey"][sl] = x + 0.5 * dx, y + 0.5 * dy
    rays = np.concatenate([rays, cross])
    after = before.copy()
    O.himm_update(g, after, rays.view(O.RAY_DTYPE))
    changed = ~((after == before) | (np.isnan(after) & np.isnan(before)))
    A, B = after.reshape(n, n), before.reshape(n, n)
    union = np.empty_like(A)
    touched_windows = 0
    for rank in range(8):
        i0, ni, j0, nj = L.window(rank)
        e.upload(R.capi.LAYER_LASER, before)
        e.himm_set_window(i0, j0, ni, nj)
        e.himm_update(R.capi.LAYER_LASER, rays)
        got = e.download(R.capi.LAYER_LASER).reshape(n, n)
        win = np.zeros((n, n), bool)
        win[j0:j0 + nj, i0:i0 + ni] = True
        assert same_f32(got[~win], B[~win]), rank                 # nothing outside the window
        union[j0:j0 + nj, i0:i0 + ni] = got[j0:j0 + nj, i0:i0 + ni]
        touched_windows += bool(changed.reshape(n, n)[win].any())
    assert same_f32(union, A)
    assert touched_windows == 8 and int(changed.sum()) > 100000
    e.close()


def test_gather_dirty_with_device_side_tile_lists():
    """dist.gather_dirty's RCCL path keeps the tile lists on the GPU (rna_last_dirty_tiles_device -> all-gather of device
    tensors -> rna_layers_unpack_tiles_device; only the counters visit the host).  RCCL needs one device per rank, so on
    the one-GPU test box the two ranks are two threads with an engine each and an in-process all-gather that hands the
    device tensors over -- the data path between the collectives is the one two GPUs take.  After the exchange both
    replicas hold the oracle's whole-map update in both layers."""
    import threading
    import torch
    import ros_navigation_amd as R
    from ros_navigation_amd import dist as D
    rows, cols, world = 300, 260, 2
    L = D.TileLayout.for_world(rows, cols, world)
    g = O.make_geom(rows * 0.05, cols * 0.05, 0.05)
    rng = np.random.default_rng(5)
    before = rng.choice(np.array([np.nan, 0, 10, 50, 150, 180], np.float32), rows * cols)
    rays = R.synth.rays(6, 700, rows * 0.05, cols * 0.05, seed=3, lmin=0.2, lmax=5.0, margin=0.5)
    after = before.copy()
    O.himm_update(g, after, rays.view(O.RAY_DTYPE))

    class ThreadDist:
        """all_gather between the threads of this process (tensors stay on the device)"""
        def __init__(self, n):
            self.n, self.slots, self.bar = n, [None] * n, threading.Barrier(n)
            self.local = threading.local()

        def get_backend(self):
            return "nccl"

        def all_gather(self, outs, t):
            torch.cuda.current_stream(t.device).synchronize()
            self.slots[self.local.rank] = t
            self.bar.wait()
            for k in range(self.n):
                outs[k].copy_(self.slots[k])
            torch.cuda.current_stream(t.device).synchronize()
            self.bar.wait()

    fake = ThreadDist(world)
    engines, errors, got_bytes = [None] * world, [], [0] * world

    def run(rank):
        try:
            fake.local.rank = rank
            e = R.Engine(rows * 0.05, cols * 0.05, 0.05)
            engines[rank] = e
            e.upload(R.capi.LAYER_LASER, before)
            e.compose_master(1)
            i0, ni, j0, nj = L.window(rank)
            e.himm_set_window(i0, j0, ni, nj)
            e.update_map(rays, compose_mode=0)
            got_bytes[rank] = D.gather_dirty(e, (R.capi.LAYER_LASER, R.capi.LAYER_MASTER), L, rank, fake)
            e.compose_master(0)
            e.synchronize()
        except Exception as ex:   # noqa: BLE001
            errors.append((rank, repr(ex)))
            fake.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert all(b > 0 and b % (64 * 64 * 4) == 0 for b in got_bytes), got_bytes
    for rank in range(world):
        e = engines[rank]
        assert same_f32(e.download(R.capi.LAYER_LASER), after), rank
        assert same_f32(e.download(R.capi.LAYER_MASTER), after), rank
        e.close()
