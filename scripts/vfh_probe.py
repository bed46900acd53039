"""Developer probe: vfh_step kernel time against the number of poses (launch floor vs per-workgroup latency)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ros_navigation_amd as R
n = 1024; L = n * 0.05
e = R.Engine(L, L, 0.05)
e.upload(R.capi.LAYER_MASTER, R.synth.occupancy_sparse(n, n, seed=1))
sizes = [int(x) for x in os.environ.get("VFH_PROBE_SIZES", "1,64,256,1024,4096,16384").split(",")]
reps = int(os.environ.get("VFH_PROBE_REPS", "20"))
for m in sizes:
    poses = R.synth.poses(m, L, L, seed=1)
    e.vfh_init(m)
    for _ in range(3): e.vfh_step(poses)
    e.profile(True); e.profile_reset()
    for _ in range(reps): e.vfh_step(poses)
    p = e.profile_get(); e.profile(False)
    print(m, round(p["vfh_step"][0] / p["vfh_step"][1] * 1e3, 1), "us")
