"""Developer probe: A* statistics on the bench map after k HIMM batches (what bench.py times)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ros_navigation_amd as R
n, nq = 4096, 256
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
e.upload(R.capi.LAYER_LASER, master)
e.compose_master(1)
rays = R.synth.rays(64, 1563, n * 0.05, n * 0.05, seed=4)
q = R.synth.astar_queries(nq, master, n, n, seed=2)
e.astar_configure(max_queries=nq, bucket_width=8000)
e.astar_pipeline_depth(1)
e.profile(True)
for k in range(0, 7):
    if k:
        e.update_map(rays, compose_mode=0)
    e.profile_reset()
    res, paths = e.astar(q, 32768)
    settled = e.astar_settled(nq)
    prof = e.profile_get()
    m = e.download(R.capi.LAYER_MASTER)
    print("himm batches=%d blocked=%.4f search=%.1f ms found=%d settled mean/max=%d/%d expanded mean/max=%d/%d rounds mean/max=%d/%d" % (
        k, float((np.nan_to_num(m) > 0).mean()), prof["astar_search"][0], int((res["status"] == 0).sum()), settled.mean(), settled.max(),
        res["expanded"].mean(), res["expanded"].max(), res["rounds"].mean(), res["rounds"].max()))
