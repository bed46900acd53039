"""RRT row alone: kernel time and the distribution of samples / tree nodes per query (config 4's per-GPU share).
With a library built with -DRNA_RRT_STATS (RNA_LIB=...) the kernel also prints its per-sample time split."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ros_navigation_amd as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.zeros(1, device="cuda")
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=3)
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.rrt_queries(nq, master, n, n, e.get_position, seed=3, max_samples=100000)
res, paths = e.rrt(q)
e.profile(True)
e.profile_reset()
res, paths = e.rrt(q)
ms = e.profile_get()["rrt"][0]
s = np.sort(res["samples"])
print("rrt %d queries on %d^2: %.1f ms | samples total %d, median %d, p90 %d, max %d (x%d at max) | nodes total %d | reached %d aborted %d" % (
    nq, n, ms, s.sum(), s[len(s) // 2], s[int(len(s) * 0.9)], s[-1], int((s == s[-1]).sum()), res["tree_size"].sum(),
    int((res["status"] == 1).sum()), int((res["status"] == -1).sum())))
print("us per sample on the longest query: %.2f" % (ms * 1e3 / s[-1]))
ab = res["status"] == -1
print("aborted queries: tree sizes", sorted(res["tree_size"][ab].tolist()))
big = np.argsort(-res["samples"])[:24]
print("the 24 queries with most samples: (samples, nodes, status)", [(int(res["samples"][k]), int(res["tree_size"][k]), int(res["status"][k])) for k in big])
if os.environ.get("RRT_ONLY_ABORTED"):
    # the longest chains by themselves: the batch's time is theirs (each alone on a CU)
    sel = np.flatnonzero(ab)
    q2 = q[sel].copy()
    print("--- the %d aborted queries alone" % len(sel), file=sys.stderr); sys.stderr.flush()
    e.profile_reset()
    r2, _ = e.rrt(q2)
    ms2 = e.profile_get()["rrt"][0]
    print("aborted alone: %.1f ms for %d samples each -> %.3f us per sample; tree sizes %s" % (ms2, int(r2["samples"].max()), ms2 * 1e3 / r2["samples"].max(), sorted(r2["tree_size"].tolist())))
    for ts in (1, None):
        k = [i for i in sel if (res["tree_size"][i] == 1) == (ts == 1)]
        if not k: continue
        print("--- aborted with tree %s" % ("of one node" if ts == 1 else "of more nodes"), file=sys.stderr); sys.stderr.flush()
        e.profile_reset()
        r3, _ = e.rrt(q[k].copy())
        ms3 = e.profile_get()["rrt"][0]
        print("  %d queries, trees %s: %.1f ms -> %.3f us per sample" % (len(k), sorted(r3["tree_size"].tolist()), ms3, ms3 * 1e3 / r3["samples"].max()))
if os.environ.get("RRT_LONGEST_ALONE"):
    # the queries with most samples, each by itself on the chip: which chain is the batch's length?
    for k in np.argsort(-res["samples"])[:int(os.environ["RRT_LONGEST_ALONE"])]:
        e.profile_reset()
        r1, _ = e.rrt(q[k:k + 1].copy())
        print("  query %3d alone: %5.1f ms  samples %6d nodes %4d status %2d" % (k, e.profile_get()["rrt"][0], int(r1["samples"][0]), int(r1["tree_size"][0]), int(r1["status"][0])))
