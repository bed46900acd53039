"""Developer tool: time-boxed random parity run of the waypoint-graph A* kernel (AStarPlanner::makePlan on arbitrary
small graphs) against the oracle's Boost.Graph-semantics restatement: lattice vertices (equal-cost routes, ties of the
closest-vertex search), parallel edges, self loops, disconnected parts, explicit and default edge weights.
usage: python scripts/fuzz_graph.py [seconds] [seed]"""
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
L = O.lib()
e = R.Engine(2.0, 2.0, 0.05)
t_end = time.time() + budget
cases = queries = found = 0
while time.time() < t_end:
    nv = int(rng.integers(1, 80))
    if rng.random() < 0.6:
        V = rng.integers(0, 8, (nv, 2)).astype(np.float64) * float(rng.choice([1.0, 0.5, 2.5]))   # lattice: ties
    else:
        V = rng.uniform(-10, 30, (nv, 2))
    ne = int(rng.integers(0, 3 * nv + 1))
    E = rng.integers(0, nv, (ne, 2)).astype(np.int32)
    W = None
    if ne and rng.random() < 0.5:
        W = rng.choice([1.0, 2.0, 0.5, 3.25, 0.0], ne).astype(np.float32) if rng.random() < 0.5 else rng.uniform(0, 10, ne).astype(np.float32)
    nq = int(rng.integers(1, 64))
    st = rng.uniform(-12, 32, (nq, 4))
    if rng.random() < 0.5:
        st = np.round(st * 2) / 2                                      # queries on the half lattice: closest-vertex ties
    plen, paths = e.graph_astar(V, E, st, edge_weight=W)
    loc = np.ascontiguousarray(V).ctypes.data_as(C.POINTER(C.c_double))
    euv = np.ascontiguousarray(E).ctypes.data_as(C.POINTER(C.c_int)) if ne else None
    wp = W.ctypes.data_as(C.POINTER(C.c_float)) if W is not None else None
    verts = (C.c_int * (nv + 2))()
    for k in range(nq):
        s = L.og_graph_closest_vertex(nv, loc, O.d2(st[k, 0], st[k, 1]))
        t = L.og_graph_closest_vertex(nv, loc, O.d2(st[k, 2], st[k, 3]))
        n = L.og_graph_astar(nv, loc, ne, euv, wp, s, t, verts, nv + 2)
        want = np.zeros((0, 2)) if n == 0 else np.vstack([st[k, :2]] + [V[verts[i]] for i in range(n)] + [st[k, 2:]])
        if plen[k] != len(want) or not np.array_equal(paths[k, :len(want)], want):
            print("MISMATCH", dict(nv=nv, ne=ne, weighted=W is not None, fuzz_seed=seed, case=cases), "query", k, st[k], "closest", s, t,
                  "gpu", plen[k], paths[k, :plen[k]].tolist(), "oracle", want.tolist())
            sys.exit(1)
        found += n > 0
    queries += nq
    cases += 1
e.close()
print("graph fuzz ok: %d graphs, %d queries (%d with a path) in %.0f s, seed %d" % (cases, queries, found, budget, seed))
