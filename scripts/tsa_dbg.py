"""Developer probe (debug build): job / iteration counts and time split of the tile A* kernel."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ros_navigation_amd import capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "librna_dbg.so")
import ros_navigation_amd as R
os.environ["RNA_ASTAR_KERNEL"] = "tile"
n = 4096
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.astar_queries(256, master, n, n, seed=2)
e.astar_pipeline_depth(1)
for bw in (8000, 32000):
    e.astar_configure(max_queries=256, bucket_width=bw)
    res, paths = e.astar(q, 32768)
    tot = res["cost"].astype(np.int64) + res["expanded"] + res["rounds"] + res["buckets"]
    for k in list(np.argsort(-tot)[:3]) + [int(np.argsort(tot)[128])]:
        r = res[k]
        print("bw=%d q%d: rounds=%d jobs=%d local_iters=%d expanded=%d | wave0: load=%.1f ms local=%.1f ms writeback=%.1f ms barrier=%.1f ms (sum %.1f)" % (
            bw, k, paths[k, 0], r["status"], r["path_len"], paths[k, 1], r["cost"] * 1e-5, r["expanded"] * 1e-5, r["rounds"] * 1e-5, r["buckets"] * 1e-5, tot[k] * 1e-5))
