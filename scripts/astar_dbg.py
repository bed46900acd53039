"""Developer probe (debug build librna_dbg.so): pops / iterations / HBM-sourced rounds of the worst queries."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ros_navigation_amd import capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "librna_dbg.so")
import ros_navigation_amd as R
n = 4096
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.astar_queries(256, master, n, n, seed=2)
for bw in (8000, 16000):
    e.astar_configure(max_queries=256, bucket_width=bw)
    res, _ = e.astar(q, 32768)
    ok = res["status"] == 0
    # debug build: path_len = rounds that read from HBM, cost = pops, buckets = iterations
    for k in np.argsort(-res["rounds"])[:4]:
        print("bw=%d q%d rounds=%d iters=%d pops=%d expanded=%d hbm_rounds=%d" % (bw, k, res["rounds"][k], res["buckets"][k], res["cost"][k], res["expanded"][k], res["path_len"][k]))
    print("bw=%d totals: rounds=%d iters=%d pops=%d expanded=%d" % (bw, res["rounds"][ok].sum(), res["buckets"][ok].sum(), res["cost"][ok].sum(), res["expanded"][ok].sum()))
