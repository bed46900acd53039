import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RNA_ASTAR_KERNEL"] = "persist"
import ros_navigation_amd as R
from ros_navigation_amd import capi
n = int(sys.argv[1]); nq = int(sys.argv[2]); seed = 3
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=seed, side=(2, max(4, n // 10)))
master[np.random.default_rng(seed).random(n * n) < 0.03] = np.nan
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.astar_queries(nq, master, n, n, seed=seed)
rng = np.random.default_rng(seed)
q["start"][:6] = rng.integers(0, n * n, 6)
q["goal"][6] = q["start"][6]
e.astar_pipeline_depth(1)
L = capi.lib()
for bw in (2828, 8000, 50000):
    e.astar_configure(max_queries=nq, bucket_width=bw)
    paths = np.zeros((nq, 65536), np.int32); res = np.zeros(nq, capi.ASTAR_RESULT_DTYPE)
    rc = L.rna_astar_batch(e.h, q.ctypes.data, nq, paths.ctypes.data, 65536, res.ctypes.data)
    bad = np.nonzero(res["status"] < 0)[0]
    print("bw", bw, "rc", rc, "n bad", len(bad), "statuses", np.unique(res["status"], return_counts=True))
    for k in bad[:8]:
        print("   q%d (start %d goal %d): (status, abort_reason, qstatus, expanded, jobs, outstanding) =" % (k, q["start"][k], q["goal"][k]), res[k])
