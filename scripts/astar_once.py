"""Developer probe: one 256-query A* launch on the bench map (for rocprofv3 PMC passes)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ros_navigation_amd as R
n = 4096
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.astar_queries(256, master, n, n, seed=2)
e.astar_pipeline_depth(1)
e.astar_configure(max_queries=256, bucket_width=8000)
res, _ = e.astar(q, 32768)
print("expanded", int(res["expanded"].sum()), "rounds max", int(res["rounds"].max()))
