"""Developer probe: per-query statistics of the grid A* kernel on the bench map (not a test)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ros_navigation_amd import capi as _capi
if os.environ.get("RNA_LIB"): _capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), os.environ["RNA_LIB"])
import ros_navigation_amd as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
widths = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2828, 8000, 32000, 128000]
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.astar_queries(nq, master, n, n, seed=2)
e.profile(True)
for bw in widths:
    e.astar_configure(max_queries=nq, bucket_width=bw)
    e.astar(q, 32768)
    times = []
    for rep in range(int(os.environ.get("REPS", "7"))):
        e.profile_reset()
        res, paths = e.astar(q, 32768)
        times.append(e.profile_get()["astar_search"][0])
    e.profile_reset()
    res, paths = e.astar(q, 32768)
    print("bw=%d search ms over %d launches: min %.1f median %.1f max %.1f" % (bw, len(times), min(times), float(np.median(times)), max(times)))
    settled = e.astar_settled(nq)
    prof = e.profile_get()
    ok = res["status"] == 0
    print("bw=%d search=%.1f ms init=%.1f ms  settled mean/max=%d/%d  expanded mean/max=%d/%d (x%.2f)  rounds mean/max=%d/%d  "
          "buckets mean/max=%d/%d  pathlen max=%d  cells/round mean=%.1f" % (
              bw, prof["astar_search"][0], prof["astar_init"][0], settled.mean(), settled.max(),
              res["expanded"].mean(), res["expanded"].max(), res["expanded"].sum() / max(1, settled.sum()),
              res["rounds"].mean(), res["rounds"].max(), res["buckets"].mean(), res["buckets"].max(),
              res["path_len"].max(), res["expanded"].sum() / max(1, res["rounds"].sum())))
    worst = np.argsort(-res["rounds"])[:3]
    for k in worst:
        print("   q%d: settled=%d expanded=%d rounds=%d buckets=%d path_len=%d" % (k, settled[k], res["expanded"][k], res["rounds"][k], res["buckets"][k], res["path_len"][k]))
