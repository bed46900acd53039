"""Per-row measurements of SURVEY.md section 8 on one MI355X (not the headline bench): each hot-path
kernel on BASELINE.json's config for it, inputs resident in HBM, HIP-event timed, with the
algorithmic-byte roofline of SURVEY.md 8d.  Writes one JSON object to stdout.

    python scripts/bench_rows.py > profiles/r01_rows.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ros_navigation_amd as R  # noqa: E402

PEAK = 8000.0  # GB/s
out = {}


def dev(a):
    return torch.from_numpy(np.frombuffer(a.tobytes(), dtype=np.uint8).copy()).cuda()


def timed(e, fn, reps):
    fn()
    e.synchronize()
    e.profile(True)
    e.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    e.synchronize()
    wall = (time.perf_counter() - t0) / reps
    prof = {k: v[0] / max(1, v[1]) for k, v in e.profile_get().items() if v[1]}
    e.profile(False)
    return wall, prof


# ---- config 2: 1024^2 grid, 72-sector VFH+, 1024 poses ---------------------------------------------
n = 1024
e = R.Engine(n * 0.05, n * 0.05, 0.05)
e.upload(R.capi.LAYER_MASTER, R.synth.occupancy_sparse(n, n, seed=1))
poses = R.synth.poses(1024, n * 0.05, n * 0.05, seed=1)
e.vfh_init(len(poses))
d_poses, d_out = dev(poses), torch.zeros(len(poses) * 16, dtype=torch.uint8, device="cuda")
wall, prof = timed(e, lambda: e.vfh_step_device(d_poses.data_ptr(), len(poses), d_out.data_ptr()), 50)
bytes_pose = 4 * 31 * 31 + 2 * 72 * 4 + 32
gbs = len(poses) * bytes_pose / (prof["vfh_step"] * 1e-3) / 1e9
out["config2_vfh"] = dict(workload="1024x1024 grid, 1024 poses, Steerer VFH params", poses_per_s=len(poses) / wall,
                          kernel_ms=prof["vfh_step"], algorithmic_bytes_per_pose=bytes_pose, achieved_gbs=gbs, frac=gbs / PEAK)
# the same kernel with a batch that fills the GPU (16 384 poses): the 1024-pose row is launch-latency bound
poses16 = R.synth.poses(16384, n * 0.05, n * 0.05, seed=2)
e.vfh_init(len(poses16))
d_p16, d_o16 = dev(poses16), torch.zeros(len(poses16) * 16, dtype=torch.uint8, device="cuda")
wall16, prof16 = timed(e, lambda: e.vfh_step_device(d_p16.data_ptr(), len(poses16), d_o16.data_ptr()), 50)
gbs16 = len(poses16) * bytes_pose / (prof16["vfh_step"] * 1e-3) / 1e9
out["vfh_16384_poses"] = dict(workload="1024x1024 grid, 16384 poses", poses_per_s=len(poses16) / wall16, kernel_ms=prof16["vfh_step"],
                              achieved_gbs=gbs16, frac=gbs16 / PEAK)
e.close()

# ---- HIMM: 100k-ray batch (config 5's ray batch) on 4096^2 ------------------------------------------
n = 4096
L = n * 0.05
e = R.Engine(L, L, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
e.upload(R.capi.LAYER_LASER, master)
# four different batches in rotation, so that the clears of a step really write (a batch re-applied to its own result
# finds most cells already at 0 and skips the store)
ray_sets = [R.synth.rays(64, 1563, L, L, seed=4 + k) for k in range(4)]
d_sets = [dev(r) for r in ray_sets]
rays, d_rays = ray_sets[0], d_sets[0]
turn = [0]


def himm_step():
    k = turn[0] % 4
    turn[0] += 1
    e.update_map_device(d_sets[k].data_ptr(), len(ray_sets[k]), 0)


for _ in range(8):
    himm_step()
wall, prof = timed(e, himm_step, 20)
alg = float(np.mean([(8 * np.hypot(r["ex"] - r["sx"], r["ey"] - r["sy"]) / 0.05 + 8 * (r["clear_end"] == 0) + 40).sum() for r in ray_sets]))
k_ms = sum(prof.get(k, 0.0) for k in ("himm_prep", "himm_raster", "himm_apply"))
out["himm_batch"] = dict(workload="100032 rays (64 origins x 1563, 1-6 m) on 4096x4096, four batches in rotation, fused compose (dirty tiles)",
                         rays_per_s=len(rays) / wall, kernels_ms=prof, algorithmic_bytes=alg,
                         achieved_gbs=alg / (k_ms * 1e-3) / 1e9, frac=alg / (k_ms * 1e-3) / 1e9 / PEAK,
                         raster_gbs=alg / (prof["himm_raster"] * 1e-3) / 1e9, raster_frac=alg / (prof["himm_raster"] * 1e-3) / 1e9 / PEAK)
wall1, prof1 = timed(e, lambda: e.update_map_device(d_rays.data_ptr(), len(rays), 1), 10)
out["himm_batch"]["compose_full_copy_ms"] = prof1.get("compose_master")
out["himm_batch"]["compose_full_copy_gbs"] = 8.0 * n * n / (prof1.get("compose_master", 1) * 1e-3) / 1e9

# ---- config 3: 4096^2, 256 A* queries (static map), unpipelined launch ------------------------------
m1 = e.download(R.capi.LAYER_MASTER)
q = R.synth.astar_queries(256, m1, n, n, seed=2)
d_q = dev(q)
d_paths = torch.zeros(256 * 32768, dtype=torch.int32, device="cuda")
d_res = torch.zeros(256 * 6, dtype=torch.int32, device="cuda")
e.astar_pipeline_depth(1)
e.astar_configure(max_queries=256)
wall, prof = timed(e, lambda: e.astar_device(d_q.data_ptr(), 256, d_paths.data_ptr(), 32768, d_res.data_ptr()), 5)
settled = e.astar_settled(256)
alg = float(settled.astype(np.int64).sum()) * 44
out["config3_astar"] = dict(workload="4096x4096, 256 queries, one launch (pipeline depth 1)", queries_per_s=256 / wall,
                            search_ms=prof["astar_search"], init_ms=prof["astar_init"], settled_cells=int(settled.sum()),
                            algorithmic_bytes=alg, achieved_gbs=alg / (prof["astar_search"] * 1e-3) / 1e9,
                            frac=alg / (prof["astar_search"] * 1e-3) / 1e9 / PEAK)
e.close()

# ---- latency: the bench's own map and query set, one batch alone, and its longest search alone -----
n = 4096
for depth in (1, 13):
    e = R.Engine(n * 0.05, n * 0.05, 0.05)
    master = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
    e.upload(R.capi.LAYER_MASTER, master)
    q = R.synth.astar_queries(256, master, n, n, seed=2)
    e.astar_pipeline_depth(depth)
    e.astar_configure(max_queries=256)
    res, _ = e.astar(q, 32768)
    e.profile(True)
    ts = []
    for _ in range(3):
        e.profile_reset()
        e.astar(q, 32768)
        ts.append(e.profile_get()["astar_search"][0])
    k = int(np.argmax(res["expanded"]))
    one = q[k:k + 1].copy()
    e.astar(one, 32768)
    t1 = []
    for _ in range(3):
        e.profile_reset()
        e.astar(one, 32768)
        t1.append(e.profile_get()["astar_search"][0])
    out["astar_latency_depth%d" % depth] = dict(
        workload="bench map (4096x4096, 30%% rectangles, seed 2), the bench's first query set; pipeline depth %d: %d wavefronts per query for the batch of 256, 16 for the lone query" % (depth, 16 if depth == 1 else 8),
        batch_of_256_alone_ms=min(ts), longest_query_alone_ms=min(t1), longest_query=k)
    e.close()

# ---- config 4: RRT, 512 queries (one GPU's share of 4096) on 2048^2 ---------------------------------
n = 2048
e = R.Engine(n * 0.05, n * 0.05, 0.05)
master = R.synth.obstacles_rect(n, n, density=0.30, seed=3)
e.upload(R.capi.LAYER_MASTER, master)
q = R.synth.rrt_queries(512, master, n, n, e.get_position, seed=3, max_samples=100000)
t0 = time.perf_counter()
res, paths = e.rrt(q)
wall = time.perf_counter() - t0
e.profile(True)
e.profile_reset()
res, paths = e.rrt(q)
prof = e.profile_get()
nodes = int(res["tree_size"].sum())
alg = float((24.0 * res["tree_size"].astype(np.float64) ** 2 / 2 + 452.0 * res["samples"]).sum())
out["config4_rrt"] = dict(workload="2048x2048, 512 queries (per-GPU share of 4096), <= 2000 nodes each",
                          queries_per_s=512 / (prof["rrt"][0] * 1e-3), kernel_ms=prof["rrt"][0], reached=int((res["status"] == 1).sum()),
                          tree_nodes=nodes, samples=int(res["samples"].sum()), algorithmic_bytes=alg,
                          achieved_gbs=alg / (prof["rrt"][0] * 1e-3) / 1e9)
e.close()
print(json.dumps(out, indent=1))
