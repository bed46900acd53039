"""Developer tool: replays a failing case that scripts/fuzz_astar.py saved (moved map, buffer content, queries) against the
oracle.  usage: python scripts/replay_astar.py case.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ros_navigation_amd as R  # noqa: E402
import _oracle as O  # noqa: E402


def replay(path, verbose=True):
    d = np.load(path)
    rows, cols = int(d["rows"]), int(d["cols"])
    e = R.Engine(rows * 0.05, cols * 0.05, 0.05)
    g = O.make_geom(rows * 0.05, cols * 0.05, 0.05)
    import ctypes as C
    dummy = np.zeros(rows * cols, np.float32)
    ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(dummy))
    regs = (O.Region * 4)()
    mv = C.c_int(0)
    target = (float(d["pos"][0]), float(d["pos"][1]))
    O.lib().og_move(C.byref(g), ptrs, 1, O.d2(*target), regs, C.byref(mv))
    e.move(*target)
    assert tuple(e.geometry().start_index) == tuple(g.start) == tuple(int(x) for x in d["start_index"]), (tuple(g.start), d["start_index"])
    ref = d["ref"].copy()
    for l in range(3):
        e.upload(l, ref)
    e.compose_master(1)
    if verbose:   # the engine's neighbour masks (buffer order) against the oracle's on the unwrapped map
        s0, s1 = int(d["start_index"][0]), int(d["start_index"][1])
        gm = e.nbr_mask().reshape(cols, rows)                       # [bj, bi]
        um = np.roll(np.roll(ref.reshape(cols, rows), -s1, axis=0), -s0, axis=1)   # map space [j, i]
        _, onbr = O.astar_masks(um.reshape(-1).copy(), rows, cols)
        onbr = onbr.reshape(cols, rows)
        gmap = np.roll(np.roll(gm, -s1, axis=0), -s0, axis=1)
        diff = np.argwhere(gmap != onbr)
        print("mask cells that differ from the oracle (map space j, i):", len(diff), [(int(j), int(i), int(gmap[j, i]), int(onbr[j, i])) for j, i in diff[:12]])
    q = d["q"]
    e.astar_configure(max_queries=len(q), bucket_width=int(d["bucket_width"]))
    res, paths = e.astar(q, rows * cols)
    settled = e.astar_settled(len(q))
    bad = []
    for k in range(len(q)):
        ores, opath = O.astar_query_on_map(g, ref, q["start"][k], q["goal"][k])
        ok = res["status"][k] == (0 if ores.status == 0 else 1)
        if ok and ores.status == 0:
            ok = (res["cost"][k] == ores.cost and res["path_len"][k] == ores.path_len and
                  np.array_equal(paths[k, :ores.path_len], opath) and settled[k] == ores.settled)
        if not ok:
            bad.append((k, tuple(res[k]), ores.status, ores.cost, ores.path_len, ores.settled))
            if verbose and ores.status == 0 and res["status"][k] == 0:
                s0, s1 = int(d["start_index"][0]), int(d["start_index"][1])

                def mapij(lin):   # buffer linear index -> map-space (i, j)
                    bi, bj = int(lin) % rows, int(lin) // rows
                    return ((bi - s0) % rows, (bj - s1) % cols)
                gp = [mapij(c) for c in paths[k, :res["path_len"][k]]]
                op = [mapij(c) for c in opath]
                print("oracle path (map space):", op)
                print("gpu    path (map space):", gp)
                # accumulated cost along both: where does the GPU path first cost more than the oracle's g at the same cell?
                def costs(p):
                    c, out = 0, [0]
                    for a, b in zip(p[:-1], p[1:]):
                        c += 1414 if (a[0] != b[0] and a[1] != b[1]) else 1000
                        out.append(c)
                    return out
                oc = dict(zip(op, costs(op)))
                for cell, c in zip(gp, costs(gp)):
                    if cell in oc and oc[cell] != c:
                        print("first cell on both paths with different g:", cell, "gpu", c, "oracle", oc[cell], "tile", (cell[0] // 64, cell[1] // 16), "in tile", (cell[0] % 64, cell[1] % 16))
                        break
    e.close()
    if verbose:
        print("replay", os.path.basename(path), "mismatches:", bad)
    return bad


if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    sys.exit(1 if replay(sys.argv[1]) else 0)
