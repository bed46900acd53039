"""Developer probe: time subsets of the bench queries (contention vs intrinsic latency)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ros_navigation_amd as R
n = 4096
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.astar_queries(256, master, n, n, seed=2)
e.profile(True)
for sel in ([222], [222] * 8, list(range(216, 224)), list(range(192, 256)), list(range(256))):
    qq = q[sel]
    e.astar_configure(max_queries=len(qq), bucket_width=8000)
    e.astar(qq, 32768)
    e.profile_reset()
    res, _ = e.astar(qq, 32768)
    prof = e.profile_get()
    print("nq=%d search=%.1f ms init=%.2f ms max_rounds=%d max_expanded=%d" % (len(qq), prof["astar_search"][0], prof["astar_init"][0], res["rounds"].max(), res["expanded"].max()))
