"""Developer tool: time-boxed random parity run of the grid A* kernels against the CPU oracle (status, cost, path,
settled count) over random map shapes (not multiples of the 32-cell tile), obstacle densities, unknown cells,
bucket widths, batch sizes and engine reuse (the lazy field reset).  usage: python scripts/fuzz_astar.py [seconds] [seed]
Exits non-zero on the first mismatch and prints the configuration that reproduces it."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402  (initialises the HIP runtime before librna.so loads)
import ros_navigation_amd as R  # noqa: E402
import _oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.zeros(1, device="cuda")
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = queries = found = moved_found = 0
while time.time() < t_end:
    rows, cols = int(rng.integers(3, 420)), int(rng.integers(3, 420))
    if rng.random() < 0.15:
        rows, cols = int(rng.integers(400, 900)), int(rng.integers(400, 900))
    density = float(rng.choice([0.0, 0.05, 0.2, 0.3, 0.45, 0.6]))
    side_hi = int(rng.integers(2, max(3, min(rows, cols) // 3 + 2)))
    cfg = dict(rows=rows, cols=cols, density=density, side_hi=side_hi)
    e = R.Engine(rows * 0.05, cols * 0.05, 0.05)
    assert (e.rows, e.cols) == (rows, cols), cfg
    for rep in range(int(rng.integers(1, 4))):         # the same engine plans on changing maps
        mseed = int(rng.integers(0, 1 << 30))
        master = R.synth.obstacles_rect(rows, cols, density=density, seed=mseed, side=(1, side_hi))
        if rng.random() < 0.5:
            master[rng.random(rows * cols) < 0.05] = np.nan
        e.upload(R.capi.LAYER_MASTER, master)
        nq = int(rng.integers(1, 70))
        bw = int(rng.choice([2828, 3000, 5000, 8000, 16000, 60000, 400000]))
        mq = int(rng.choice([nq, max(1, nq // 3), 256]))
        q = np.zeros(nq, R.capi.ASTAR_QUERY_DTYPE)
        free = np.flatnonzero(~(np.isfinite(master) & (master > 0)))
        if len(free) and rng.random() < 0.8:
            q["start"], q["goal"] = rng.choice(free, nq), rng.choice(free, nq)
        else:
            q["start"], q["goal"] = rng.integers(0, rows * cols, nq), rng.integers(0, rows * cols, nq)
        here = dict(cfg, mseed=mseed, nq=nq, bucket_width=bw, max_queries=mq, rep=rep, fuzz_seed=seed, case=cases)
        e.astar_configure(max_queries=mq, bucket_width=bw)
        res, paths = e.astar(q, rows * cols)
        settled = e.astar_settled(nq) if mq >= nq else None
        _, nbr = O.astar_masks(master, rows, cols)
        assert np.array_equal(e.nbr_mask(), nbr), here
        gw = np.empty(rows * cols, np.int32)
        for k in range(nq):
            ores, opath, _ = O.astar_query(nbr, rows, cols, q["start"][k], q["goal"][k], g_work=gw)
            ok = res["status"][k] == ores.status
            if ok and ores.status == 0:
                ok = (res["cost"][k] == ores.cost and res["path_len"][k] == ores.path_len and
                      np.array_equal(paths[k, :ores.path_len], opath) and (settled is None or settled[k] == ores.settled or settled[k] == -1))
                found += 1
            if not ok:
                print("MISMATCH", here, "query", k, int(q["start"][k]), int(q["goal"][k]), "gpu", res[k], "oracle", ores.status,
                      ores.cost, ores.path_len, ores.settled)
                sys.exit(1)
        queries += nq
        cases += 1
    if rng.random() < 0.5:
        # the same engine after GridMap::move: the search runs in map space, indices at the boundary are buffer indices
        g = O.make_geom(rows * 0.05, cols * 0.05, 0.05)
        ref = master.copy()
        for l in range(3):
            e.upload(l, ref)
        ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
        regs = (O.Region * 4)()
        mv = C.c_int(0)
        for step in range(int(rng.integers(1, 3))):
            target = (float(g.pos[0] + rng.uniform(-0.4, 0.4) * rows * 0.05), float(g.pos[1] + rng.uniform(-0.4, 0.4) * cols * 0.05))
            O.lib().og_move(C.byref(g), ptrs, 1, O.d2(*target), regs, C.byref(mv))
            e.move(*target)
        e.compose_master(1)
        assert tuple(e.geometry().start_index) == tuple(g.start), cfg
        nq = int(rng.integers(1, 50))
        q = np.zeros(nq, R.capi.ASTAR_QUERY_DTYPE)
        free = np.flatnonzero(~(np.isfinite(ref) & (ref > 0)))
        q["start"], q["goal"] = rng.choice(free, nq), rng.choice(free, nq)
        bw = int(rng.choice([2828, 8000, 16000, 60000]))
        here = dict(cfg, mseed=mseed, moved_to=target, start_index=tuple(g.start), nq=nq, bucket_width=bw, fuzz_seed=seed, case=cases)
        e.astar_configure(max_queries=nq, bucket_width=bw)
        res, paths = e.astar(q, rows * cols)
        settled = e.astar_settled(nq)
        for k in range(nq):
            ores, opath = O.astar_query_on_map(g, ref, q["start"][k], q["goal"][k])
            ok = res["status"][k] == (0 if ores.status == 0 else 1)
            if ok and ores.status == 0:
                ok = (res["cost"][k] == ores.cost and res["path_len"][k] == ores.path_len and
                      np.array_equal(paths[k, :ores.path_len], opath) and (settled[k] == ores.settled or settled[k] == -1))
                moved_found += 1
            if not ok:
                print("MISMATCH (moved map)", here, "query", k, int(q["start"][k]), int(q["goal"][k]), "gpu", res[k], "oracle",
                      ores.status, ores.cost, ores.path_len, ores.settled)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)   # everything scripts/replay_astar.py needs
                np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_astar_fail.npz"), ref=ref, rows=rows, cols=cols, start_index=np.array(g.start),
                         pos=np.array(g.pos), q=q, bucket_width=bw, k=k)
                sys.exit(1)
        queries += nq
    e.close()
print("fuzz ok (%d paths on moved maps): %d maps, %d queries (%d with a path) in %.0f s, seed %d" % (moved_found, cases, queries, found, budget, seed))
