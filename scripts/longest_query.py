"""Developer probe: the longest searches of the bench's query set, each alone on the GPU (batches of <= 32 queries run 16
wavefronts per query on any engine), and the whole batch of 256 at pipeline depth 1 (16 wavefronts per query) and on a
pipelined engine (8).  usage: python scripts/longest_query.py [grid] [how many]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ros_navigation_amd as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
top = int(sys.argv[2]) if len(sys.argv) > 2 else 4
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
q = R.synth.astar_queries(256, master, n, n, seed=2)
for depth in (1, 2):
    e = R.Engine(n * 0.05, n * 0.05, 0.05)
    e.upload(R.capi.LAYER_MASTER, master)
    e.astar_pipeline_depth(depth)
    e.astar_configure(max_queries=256, bucket_width=96000)
    res, _ = e.astar(q, 32768)
    order = np.argsort(-res["expanded"])[:top]
    e.profile(True)
    for k in order:
        one = q[k:k + 1].copy()
        e.astar(one, 32768)
        ts = []
        for _ in range(3):
            e.profile_reset()
            r1, _ = e.astar(one, 32768)
            ts.append(e.profile_get()["astar_search"][0])
        print("depth %d (%d wavefronts per query): query %3d expanded %8d jobs/wave %5d buckets %3d  alone %.2f ms" % (
            depth, 16, k, res["expanded"][k], r1["rounds"][0], r1["buckets"][0], min(ts)))
    e.profile_reset()
    e.astar(q, 32768)
    print("depth %d (%d wavefronts per query): the whole batch %.2f ms" % (depth, 16 if depth == 1 else 8, e.profile_get()["astar_search"][0]))
    e.close()
