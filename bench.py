#!/usr/bin/env python3
"""bench.py -- replan cycles/sec (HIMM + VFH+ + A*) on a 4096 x 4096 grid, BASELINE.json's metric.

One "step" = one TURN of the A* pipeline (--pipeline map updates, default 20), i.e. 20 passes of
    HIMM ray batch (64 robot origins x 1563 rays) on the laser layer + fused compose-master
    -> VFH+ step for 256 robot poses -> grid A* for 256 (start, goal) queries
all resident in HBM: 20 x 256 = 5120 replan cycles per step.  The searches of 20 consecutive passes are in
flight at once (each on its own pipeline stage), so a single pass is not a unit whose time can be
measured by itself: a step hands every stage one batch, and the driver's 20 steps then time 400
passes of steady state instead of 20 passes through a pipeline that is empty at both ends.
A "replan cycle" is one (pose -> VFH command, start/goal -> A* path) pair served against the map
that has received its HIMM batch; the ray batch is amortised over the 256 cycles of its pass
(SURVEY.md section 8d).  The passes rotate through ROTATE pre-generated ray batches, pose sets and
query sets, so no pass repeats its predecessor's input (HIMM clears really write, every A* batch is
a different one).

N > 1 (`--gpus N`): one process per GPU, each with its own replicated grid and its own shard of
poses/queries (weak scaling, no data-path collective); value = all ranks' cycles / max time.  Started
without torchrun, the parent spawns the N ranks itself (before anything touches a GPU) and exits with
their status; under `torch.distributed.run` (RANK/WORLD_SIZE set) WORLD_SIZE must equal --gpus.

Prints ONE JSON line on rank 0 (see the task contract) including
  roofline:      dominant kernel (astar_search) algorithmic GB/s vs the 8 TB/s HBM peak
  roofline_rows: the same for himm_raster and vfh_step, from the same run
  cpu_baseline:  the CPU oracle ("port") timed on a bounded sample of the same workload
"""
import argparse
import json
import os
import subprocess
import sys
import time

# pipelined A* batches run on several HIP streams; give the runtime enough hardware queues
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
DEFAULT_PIPELINE = 20   # A* batches in flight on one GPU (the committed counter files under profiles/ belong to it).  Round 6, default
                        # bench / the driver's 20-step run: 16 -> 162.0 / 158.0 k, 18 -> 162.5 / 161.5 k, 20 -> 164.0 / 162.5 k cycles/s
                        # (profiles/r06_sweep_depth.txt, r06_sweep_depth_final.txt: three runs each on two boxes, 20 ahead of 18 in every
                        # one); from 22 on the rate falls to 135 k whatever GPU_MAX_HW_QUEUES is (the process holds 25 queues then: r06_sweep_depth_hw_queues.txt) and
                        # the engine takes 20 at most
# ... and next to an RCCL communicator, whose streams take hardware queues of their own: the rate falls off from 21 stages on
# without one and from 18 on with one (profiles/r04_sweep_depth_rccl.txt); 14 and 16 measure the same there
DEFAULT_PIPELINE_RCCL = 14

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
ASTAR_BYTES_PER_SETTLED = 44   # 8 neighbour occupancy reads x 4 B + 12 B g/parent/flag RMW
VFH_BYTES_PER_POSE = 4 * 31 * 31 + 2 * 72 * 4 + 32
ROTATE = int(os.environ.get("RNA_BENCH_ROTATE", "4"))   # distinct ray batches / pose sets / query sets the steps cycle through (developer override)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=54, help="timed steps; one step = one turn of the pipeline = --pipeline passes of "
                                                           "[HIMM batch, VFH+ x queries, A* x queries] (default: 972 passes)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed steps (turns of the pipeline)")
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--queries", type=int, default=256, help="A* queries == VFH poses per step (cycles per step)")
    ap.add_argument("--ray-poses", type=int, default=64)
    ap.add_argument("--rays-per-pose", type=int, default=1563)
    ap.add_argument("--bucket-width", type=int, default=0)
    ap.add_argument("--queue-capacity", type=int, default=0)
    ap.add_argument("--max-path", type=int, default=32768)
    ap.add_argument("--pipeline", type=int, default=0,
                    help="A* batches in flight (rna_astar_set_pipeline_depth); 0 = 20 on one GPU, 14 next to an RCCL communicator.  "
                         "Four search workgroups share a CU, so ~900 queries "
                         "run at once and the batches' tails differ (round 4, default steps / the driver's 20: 13: 148.9k / 146.5k, "
                         "16: 151.1k / 147.9k cycles/s -- profiles/r04_sweep_depth_13_16.txt; round 6: profiles/r06_sweep_depth.txt; 20 is the engine's maximum, and more than two stages need "
                         "GPU_MAX_HW_QUEUES=8, set above: with the runtime's default of 4 queues the rate collapses).  "
                         "Every launch is stretched by the ones it overlaps with, and "
                         "the roofline line divides by that per-launch duration")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-check-paths", action="store_true", help="skip the path check that is the default on one GPU")
    ap.add_argument("--check-paths", action="store_true",
                    help="(default at --gpus 1, so that the driver's own record is path-certified; about 2 s on the box's host "
                         "cores, outside the timed region) after the timed region: the LAST batch of one more turn of the pipeline (searched while the turn's other "
                         "batches are in flight) is handed to the CPU oracle -- status, cost, length and every cell of every path "
                         "of its queries against the map that batch saw (`config.paths_checked`; a mismatch ends the run)")
    ap.add_argument("--tiled-full-gather", action="store_true", help="--tiled: all-gather whole windows instead of dirty tiles")
    ap.add_argument("--tiled", action="store_true",
                    help="SURVEY 8e mode 2 / BASELINE config 5: ONE map tiled 2 x N/2 over the GPUs (windowed HIMM, halo "
                         "exchange for VFH+, all-gather of the owner windows for A*) instead of replicated maps")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process would see, counted WITHOUT opening the HIP runtime (the parent of the ranks must never touch a
    GPU: a process that has may not start children that exec): compute nodes of the KFD topology (`simd_count` > 0),
    narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None if the topology is unreadable."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def pin_to_gpu_numa_node(local_rank):
    """Confine this process (and every thread it starts later: the engine's launch thread, torch's, RCCL's proxy) to the
    host cores of the NUMA node its GPU hangs on -- BEFORE torch or HIP are loaded.  Eight ranks share one host at N = 8:
    un-pinned, a rank's launch thread and its GPU may sit on different sockets and every doorbell / pinned-memory flag
    crosses the inter-socket link.  The GPU of rank r is the r-th compute node of the KFD topology (HIP's order), narrowed
    by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when they are plain indices; its NUMA node is read from the DRM device
    behind the node's render minor.  RNA_BENCH_CPUS=n (developer / test switch) keeps only the first n of those cores.
    Returns what was done for `config.host_affinity`; never fails (an unreadable topology leaves the affinity alone)."""
    import glob
    info = {"numa_node": None, "cpus": len(os.sched_getaffinity(0)), "pinned": False}
    try:
        cpus = set(os.sched_getaffinity(0))
        before = set(cpus)
        _AFFINITY_AT_START[:] = sorted(before)
        if os.environ.get("RNA_BENCH_NO_PIN") != "1":
            gpus = []
            for path in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"), key=lambda p: int(p.split("/")[-2])):
                with open(path) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    gpus.append(int(props.get("drm_render_minor", "-1")))
            for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
                v = os.environ.get(var)
                if v is not None and all(x.strip().isdigit() for x in v.split(",") if x.strip()):
                    gpus = [gpus[int(x)] for x in v.split(",") if x.strip() and int(x) < len(gpus)]
            if 0 <= local_rank < len(gpus) and gpus[local_rank] >= 0:
                with open("/sys/class/drm/renderD%d/device/numa_node" % gpus[local_rank]) as f:
                    node = int(f.read().strip())
                info["numa_node"] = node
                if node >= 0:
                    with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                        want = set()
                        for part in f.read().strip().split(","):
                            a, _, b = part.partition("-")
                            want.update(range(int(a), int(b or a) + 1))
                    if cpus & want:
                        cpus &= want
                        info["pinned"] = True
        keep = int(os.environ.get("RNA_BENCH_CPUS", "0"))
        if keep > 0:
            cpus = set(sorted(cpus)[:keep])
        if cpus != before:
            os.sched_setaffinity(0, cpus)    # (refused with EPERM inside some sandboxes: reported, not fatal)
        info["cpus"] = len(os.sched_getaffinity(0))
    except Exception as ex:   # noqa: BLE001 -- an optimisation, never a reason to fail the bench
        info["error"] = repr(ex)
        info["pinned"] = False
    return info


_AFFINITY_AT_START = []   # the cores the process started with (the CPU baseline leg goes back to them: it is timed on all host cores)


def spawn_ranks(args):
    """`--gpus N` without a launcher: start the N ranks as fresh child processes (this process has not touched a GPU
    and never will: it imports neither torch nor the engine) and exit with their status."""
    share = os.environ.get("RNA_BENCH_SHARE_GPU") == "1"
    if not share:
        have = visible_gpus()
        if have is not None and have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
        # (unreadable topology: every rank checks its own device and exits non-zero without one)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # what the ranks report about their parent (`config.launcher`): it has loaded neither torch nor the engine, and no
    # HIP / HSA runtime is mapped into it
    with open("/proc/self/maps") as f:
        maps = f.read()
    hip_free = not any(m in sys.modules for m in ("torch", "ros_navigation_amd")) and "libamdhip64" not in maps and "libhsa-runtime" not in maps
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   RNA_BENCH_PARENT="%d:%d" % (os.getpid(), 1 if hip_free else 0))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def check_paths(master, queries, res, paths, rows, cols):
    """The checker leg of --check-paths: one batch of the loop against the oracle (the CPU restatement of the search contract,
    tests/_oracle.py -> oracle/astar.c) on the map that batch saw.  Not part of any timed region."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from concurrent.futures import ThreadPoolExecutor
    _, nbr = O.astar_masks(master, rows, cols)
    cores = max(1, len(os.sched_getaffinity(0)))

    def one(k):
        ores, opath, _ = O.astar_query(nbr, rows, cols, queries["start"][k], queries["goal"][k], path_cap=rows * cols)
        if ores.status != 0:
            return int(res[k, 0] == 1), 0
        ok = res[k, 0] == 0 and res[k, 1] == ores.path_len and res[k, 2] == ores.cost and np.array_equal(paths[k, :ores.path_len], opath)
        return int(ok), 1
    with ThreadPoolExecutor(min(cores, 32)) as ex:
        got = list(ex.map(one, range(len(queries))))
    return {"queries": len(queries), "matched": sum(g[0] for g in got), "paths_found": sum(g[1] for g in got),
            "against": "oracle/astar.c on the master layer as the batch's mask snapshot saw it (the last pass of a full turn of the "
                       "pipeline, i.e. searched next to the turn's other batches)"}


def cpu_baseline(args, R, master, rays, poses, queries, rows, cols, length):
    """Oracle (CPU restatement of the reference path) on a bounded sample: 1 thread, then one thread per host core."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    g = O.make_geom(length, length, 0.05)
    laser = master.copy()
    t0 = time.perf_counter()
    O.himm_update(g, laser, rays.view(O.RAY_DTYPE))
    mast = laser.copy()  # compose: master = laser (whole-layer copy, as the reference does)
    t_himm = time.perf_counter() - t0
    n_vfh = min(len(poses), 256)
    vs = [O.OracleVfh() for _ in range(n_vfh)]
    t0 = time.perf_counter()
    for k in range(n_vfh):
        p = poses[k]
        vs[k].step_pose(g, mast, p["x"], p["y"], p["yaw"], int(p["current_speed"]), p["goal_direction"],
                        p["goal_distance"], p["goal_tolerance"], float(p["dt"]))
    t_vfh = (time.perf_counter() - t0) / n_vfh
    _, nbr = O.astar_masks(mast, rows, cols)
    gw = np.empty(rows * cols, np.int32)
    t_a, n_a, settled, settled_goal = 0.0, 0, 0, 0
    while n_a < len(queries) and (n_a < 2 or t_a < args.cpu_seconds):
        q = queries[n_a]
        t0 = time.perf_counter()
        res, _, _ = O.astar_query(nbr, rows, cols, q["start"], q["goal"], path_cap=rows * cols, g_work=gw)
        t_a += time.perf_counter() - t0
        settled += res.settled
        settled_goal += O.astar_last_settled_at_goal() if res.status == 0 else res.settled
        n_a += 1
    # the one reference translation unit that compiles here (move_control's own vfh.cpp -> oracle/_ref, built by oracle/Makefile
    # where /root/reference exists and shipped to the GPU box as a binary): its Update_VFH timed beside the port's, on the
    # port's own ranges (getRangesFromSubmap is in steerer.cpp, which needs ROS and cannot be compiled)
    vfh_ref_us = vfh_port_us = None
    try:
        if O.ref() is not None:
            rngs = [O.ranges_array(O.ranges_from_submap(g, mast, poses[k]["x"], poses[k]["y"], poses[k]["yaw"])[1]) for k in range(n_vfh)]
            for cls in (O.RefVfh, O.OracleVfh):
                inst = [cls() for _ in range(n_vfh)]
                t0 = time.perf_counter()
                for k in range(n_vfh):
                    p = poses[k]
                    if cls is O.RefVfh:
                        O.ref().refvfh_set_clock(1000.0)   # (the shim's clock is one per process: every instance was created at 1000.0)
                    inst[k].update(rngs[k], int(p["current_speed"]), float(p["goal_direction"]), float(p["goal_distance"]),
                                   float(p["goal_tolerance"]), float(p["dt"]))
                us = (time.perf_counter() - t0) / n_vfh * 1e6
                vfh_ref_us, vfh_port_us = (us, vfh_port_us) if cls is O.RefVfh else (vfh_ref_us, us)
    except Exception:   # noqa: BLE001 -- a side figure: the baseline stands without it
        vfh_ref_us = vfh_port_us = None
    per_cycle_1 = t_himm / len(queries) + t_vfh + t_a / n_a
    # (ii) one thread per host core over independent A* queries (ctypes releases the GIL inside the C oracle) -- on ALL the cores
    # the process started with, not only those of the GPU's NUMA node the timed region was pinned to
    if _AFFINITY_AT_START:
        try:
            os.sched_setaffinity(0, set(_AFFINITY_AT_START))
        except OSError:
            pass
    cores = max(1, len(os.sched_getaffinity(0)))
    from concurrent.futures import ThreadPoolExecutor
    n_mt = min(len(queries), max(cores, int(round(cores * args.cpu_seconds / max(t_a / n_a, 1e-6)))))

    def run_slice(w):
        gwork = np.empty(rows * cols, np.int32)
        for k in range(w, n_mt, cores):
            O.astar_query(nbr, rows, cols, queries[k]["start"], queries[k]["goal"], path_cap=rows * cols, g_work=gwork)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(run_slice, range(cores)))
    t_mt = (time.perf_counter() - t0) / n_mt
    # HIMM (ray order matters) and VFH (32 us per pose: not worth a thread hand-over) are charged at their measured
    # single-thread cost in both figures; only the A* queries are spread over the cores
    per_cycle = t_himm / len(queries) + t_vfh + t_mt
    return {"value": 1.0 / per_cycle, "unit": "replan cycles/s", "cores": cores, "kind": "port",
            "value_1core": 1.0 / per_cycle_1,
            # Update_VFH alone (ranges given), per pose on one core: the reference's own vfh.cpp (oracle/_ref) and the port
            "vfh_reference_us": vfh_ref_us, "vfh_port_update_us": vfh_port_us,
            # E both ways on the single-thread sample: the contract settles the whole f == f* plateau (canonical ties), an A*
            # that stops when the goal comes off the heap closes fewer cells
            "settled_cells_contract": settled / n_a, "settled_cells_stop_at_goal": settled_goal / n_a,
            "sample": "oracle (C, -O2) on the first of the %d rotating input sets: full %d-ray HIMM batch + compose (1 thread, "
                      "%.3f s, amortised over %d cycles), %d VFH+ poses (%.1f us each on one core), A*: first %d queries on 1 "
                      "thread (%.3f s each, %.0f cells settled each) and first %d queries on %d threads (%.4f s per query wall; "
                      "the oracle refills a %d MB g[] per query -- oracle/astar.c -- so the many-thread figure is bound by "
                      "the host's memory bandwidth, not by its cores)"
                      % (ROTATE, len(rays), t_himm, len(queries), n_vfh, t_vfh * 1e6, n_a, t_a / n_a, settled / n_a, n_mt,
                         cores, t_mt, rows * cols * 4 >> 20)}


# the kernels timed as the "himm_raster" slot (himm.hip: rays binned to 64 x 64 tiles, then rasterised per tile in LDS);
# the 16 KiB memset of the bin counters in front of them is not attributed
HIMM_RASTER_CHAIN = ["himm_bin_count_kernel", "himm_bin_scan_kernel", "himm_bin_fill_kernel", "himm_tile_raster_kernel"]


def pmc_traffic(kernel, args, world):
    """HBM bytes per launch of `kernel` (or of a chain of kernels timed as one slot) from the committed rocprofv3 PMC passes, as the range [TCC_EA0 requests x 64 B,
    reads doubled] (MI355X_MICROARCH.md: FETCH_SIZE reports half of the bytes of 16 B/lane streaming reads; this
    kernel's dword accesses are uncalibrated).  None unless the passes were taken on this very configuration."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        c = d["config"]
        if (c["grid"], c["queries"], c["pipeline"], c["ray_poses"], c["rays_per_pose"]) != \
                (args.grid, args.queries, args.pipeline, args.ray_poses, args.rays_per_pose) or world != 1 or args.tiled:
            return None, "PMC passes in profiles/ were taken on another configuration"
        def find(name):   # template instantiations carry their arguments in the profiler's kernel name
            hits = [v for k, v in d["kernels"].items() if k == name or k.startswith(name + "<")]
            return max(hits, key=lambda v: v["hbm_bytes_per_launch"])
        ks = [find(name) for name in ([kernel] if isinstance(kernel, str) else kernel)]   # a slot's chain: one launch each
        lo = sum(k["fetch_size_kb_avg"] + k["write_size_kb_avg"] for k in ks) * 1024.0
        hi = sum(2.0 * k["fetch_size_kb_avg"] + k["write_size_kb_avg"] for k in ks) * 1024.0
        return [lo, hi], "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; " \
                         "[(FETCH+WRITE), (2*FETCH+WRITE)] x 1024 B per launch" % os.path.relpath(PMC_SUMMARY, ROOT)
    except Exception:
        return None, "no PMC summary committed for this kernel"


SQ_COUNTERS = os.path.join(ROOT, "profiles", "r06_search_sq_counters.txt")
VALU_PEAK_PER_NS_SIMD = 0.58      # profiles/r03_ubench_valu.txt: eight wavefronts per SIMD issue 0.54-0.59 dependent VALU instructions per ns
SEARCH_SIMDS = (256 - 32) * 4     # the search streams' CU mask leaves 32 of the 256 CUs to the engine stream


def valu_issue(args, world, wall_per_pass):
    """What the search kernel is really bound by: wavefront VALU instructions of one 256-query batch, from
    the committed SQ counter pass of this workload's first query set, over the pass wall time and the SIMDs the searches
    may use.  None unless the counters were taken on this configuration."""
    try:
        if (args.grid, args.queries, args.pipeline) != (4096, 256, DEFAULT_PIPELINE) or world != 1 or args.tiled:
            return None
        valu = salu = None
        for line in open(SQ_COUNTERS):
            f = line.split()
            if len(f) >= 3 and f[0] == "SQ_INSTS_VALU":
                valu = float(f[-1].split("=")[-1])
            if len(f) >= 3 and f[0] == "SQ_INSTS_SALU":
                salu = float(f[-1].split("=")[-1])
        per_ns_simd = valu / (wall_per_pass * 1e9) / SEARCH_SIMDS
        return {"bound": "valu issue", "achieved": per_ns_simd, "peak": VALU_PEAK_PER_NS_SIMD, "unit": "wavefront VALU instructions / ns / SIMD",
                "frac": per_ns_simd / VALU_PEAK_PER_NS_SIMD, "valu_per_batch": valu, "salu_per_batch": salu,
                "source": "%s (one batch alone), profiles/r03_ubench_valu.txt (the SIMD's issue rate); "
                          "224 CUs x 4 SIMDs" % os.path.relpath(SQ_COUNTERS, ROOT), "provenance": profile_provenance(SQ_COUNTERS)}
    except Exception:
        return None


def kernel_source_sha():
    """sha256[:12] of the search kernel's source: the profile files under profiles/ carry the value they were taken at, so a
    line that quotes them says by itself whether they describe the kernel that ran (`stale`)."""
    import hashlib
    with open(os.path.join(ROOT, "ros_navigation_amd", "csrc", "astar_tile.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:12]


def profile_provenance(path):
    """{"file", "kernel_source_sha", "stale"} of a committed profile file: the sha is read from a `kernel_source_sha` line / key in
    the file (scripts/profile_r06.sh writes it), None for files of earlier rounds."""
    import re
    sha = None
    try:
        m = re.search(r"kernel_source_sha\W+([0-9a-f]{12})", open(path).read())
        sha = m.group(1) if m else None
    except OSError:
        pass
    return {"file": os.path.relpath(path, ROOT), "kernel_source_sha": sha, "stale": sha != kernel_source_sha()}


def work_inflation(args, world, settled_per_launch, counters, passes):
    """How much work the search kernel does per cell the oracle settles.  The job figures are OBSERVED IN THIS RUN: the search
    kernels count their jobs (rna_astar_job_counters, seven atomic adds per search) and the timed region's totals are read after
    it -- tile jobs per touched tile, the share of jobs that find nothing better in their halo, sticky turns, rows written.
    (Rounds 4-5 parsed these from the committed output of a -DRNA_TSA_STATS build, whose wall-clock timers slow the kernel down
    by a factor of 2-3 and are not quoted any more.)  Instructions per settled cell still come from a committed SQ counter pass of
    one batch alone (counters need rocprofv3), with the kernel source hash it was taken at."""
    out = {}
    try:
        c = counters
        if c and c["searches"] > 0 and c["jobs"] > 0:
            out.update({
                "observed_in": "the timed region of this run (rna_astar_job_counters)",
                "searches": c["searches"], "tile_jobs": c["jobs"],
                "tiles_touched_per_search": c["tiles_touched"] / c["searches"],
                "jobs_per_search": c["jobs"] / c["searches"],
                "jobs_per_touched_tile": c["jobs"] / max(1, c["tiles_touched"]),
                "noop_job_frac": c["jobs_noop"] / c["jobs"],
                "sticky_turn_frac": c["sticky_turns"] / c["jobs"],
                "rows_written_per_job": c["rows_written"] / c["jobs"],
                "buckets_per_search": c["buckets"] / c["searches"],
                # (a -DRNA_TSA_IDLE developer build reports wavefront life / idle ticks in the last two counters instead)
                "bucket_reruns_per_search": c["bucket_reruns"] / c["searches"],
                "idle_frac_developer_build": (c["bucket_reruns"] / max(1, c["buckets"])) if os.environ.get("RNA_LIB", "").startswith("librna_idle") else None,
                # a tile holds 1024 cells: jobs per tile's worth of settled cells (the verdict's "revisit factor")
                "jobs_per_1024_settled_cells": c["jobs"] / max(1.0, settled_per_launch * passes / 1024.0),
                "cells_written_per_settled_cell": 64.0 * c["rows_written"] / max(1.0, settled_per_launch * passes)})
        if (args.grid, args.queries, args.pipeline) == (4096, 256, DEFAULT_PIPELINE) and world == 1 and not args.tiled:
            valu = salu = settled = None
            for line in open(SQ_COUNTERS):
                f = line.split()
                if len(f) >= 3 and f[0] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
                    v = float(f[-1].split("=")[-1])
                    valu, salu = (v, salu) if f[0] == "SQ_INSTS_VALU" else (valu, v)
                if len(f) >= 2 and f[0] == "SETTLED_CELLS_OF_THE_BATCH":
                    settled = float(f[1])
            out.update({"instructions_per_settled_cell": (valu + salu) / settled, "valu_per_settled_cell": valu / settled,
                        "salu_per_settled_cell": salu / settled,
                        "instructions_source": "SQ_INSTS_VALU + SQ_INSTS_SALU of one 256-query batch alone (the first query set on the untouched "
                                               "bench map) / the cells the oracle settles for that batch",
                        "instructions_provenance": profile_provenance(SQ_COUNTERS)})
            if out.get("tile_jobs"):
                jobs_per_batch = out["tile_jobs"] / passes
                out["valu_per_job"] = valu / jobs_per_batch
                out["salu_per_job"] = salu / jobs_per_batch
        return out or None
    except Exception:
        return out or None


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # developer switch: RNA_BENCH_FORCE_SPAWN=1 takes the spawn path (parent -> Popen -> rank -> RCCL init -> barriers ->
    # max over ranks) with a single rank too, so that the code an N-GPU run takes can run on a one-GPU box
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("RNA_BENCH_FORCE_SPAWN") == "1"):
        if args.gpus == 1:
            os.environ["RNA_BENCH_FORCE_DIST"] = "1"
        spawn_ranks(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:   # (before the heavy imports: a mis-launch fails in milliseconds)
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s (launch with --nproc-per-node %d or drop the launcher)"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "1"), args.gpus))
    # stdout is for the ONE JSON line: libraries that write to file descriptor 1 themselves (RCCL prints a five-line
    # version banner there through C stdio, flushed when the process exits, i.e. AFTER the line) get stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    share_gpu = os.environ.get("RNA_BENCH_SHARE_GPU") == "1"
    host_affinity = pin_to_gpu_numa_node(0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0")))   # before torch / HIP start their threads
    import numpy as np
    import torch
    from ros_navigation_amd import capi as _capi
    if os.environ.get("RNA_LIB"):  # developer switch: alternative build of librna.so
        _capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), os.environ["RNA_LIB"])
    import ros_navigation_amd as R
    from ros_navigation_amd import dist as D
    rank, local_rank, world = D.env_rank_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d or drop the launcher)"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    # developer switch: RNA_BENCH_SHARE_GPU=1 runs every rank on cuda:0 with a gloo barrier, to exercise the
    # multi-rank code path on a one-GPU box (RCCL refuses two ranks on one device)
    share = os.environ.get("RNA_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if args.pipeline <= 0:
        args.pipeline = DEFAULT_PIPELINE if (world == 1 and os.environ.get("RNA_BENCH_FORCE_DIST") != "1") else DEFAULT_PIPELINE_RCCL
    # developer switch: RNA_BENCH_FORCE_DIST=1 initialises RCCL and runs the barriers with a single rank too (how many
    # hardware queues are left for the search streams once a communicator exists can then be measured on a one-GPU box)
    use_dist = world > 1 or os.environ.get("RNA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist = D.init("gloo") if share else D.init("nccl", torch.device("cuda", local_rank))

    n = args.grid
    length = n * 0.05
    e = R.Engine(length, length, 0.05, device=local_rank)
    assert (e.rows, e.cols) == (n, n)
    # map: config 3's obstacle field (seed 2) is the laser layer the HIMM batches work on
    master0 = R.synth.obstacles_rect(n, n, density=0.30, seed=2)
    e.upload(R.capi.LAYER_LASER, master0)
    e.compose_master(1)
    # ROTATE ray batches (different origins and bearings; the same on every replica).  Untimed setup: the map receives
    # every batch a few times, so that the (start, goal) pairs are drawn from cells of the map the timed steps work on
    ray_sets = [R.synth.rays(args.ray_poses, args.rays_per_pose, length, length, seed=4 + k) for k in range(ROTATE)]
    for _ in range(3):
        for rays in ray_sets:
            e.update_map(rays, compose_mode=0)
    master = e.download(R.capi.LAYER_MASTER)
    # cells a ray of any batch ends on are marked and cleared over and over: keep query end points off them
    avoid = master.copy()
    for rays in ray_sets:
        for x, y in ((rays["ex"], rays["ey"]), (rays["sx"], rays["sy"])):
            i = np.floor((length / 2 - x) / 0.05).astype(np.int64)
            j = np.floor((length / 2 - y) / 0.05).astype(np.int64)
            ok = (i >= 1) & (j >= 1) & (i < n - 1) & (j < n - 1)
            for di in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    avoid[(j[ok] + dj) * n + i[ok] + di] = 180.0
    # weak scaling: the global batch holds `queries` cycles per GPU; each rank serves its own shard
    lo, hi = D.shard_bounds(args.queries * world, rank, world)
    nq = hi - lo
    pose_sets = [R.synth.poses(args.queries * world, length, length, seed=1 + k)[lo:hi] for k in range(ROTATE)]
    layout = halo = None
    if args.tiled:
        # one map, one window per GPU: a pose is served by the GPU that owns its cell
        layout = D.TileLayout.for_world(n, n, world)
        halo = D.vfh_halo(0.05)
        e.himm_set_window(*[layout.window(rank)[k] for k in (0, 2, 1, 3)])
        pose_sets = []
        for k in range(ROTATE):
            cand = R.synth.poses(args.queries * world * 4, length, length, seed=1 + k)
            idx = np.array([e.get_index(p["x"], p["y"]) for p in cand])
            own = cand[layout.owner(idx[:, 0], idx[:, 1]) == rank][:nq].copy()
            assert len(own) == nq, "not enough synthetic poses inside this rank's window"
            pose_sets.append(own)
    query_sets = [R.synth.astar_queries(args.queries * world, avoid, n, n, seed=2 + k)[lo:hi] for k in range(ROTATE)]

    dev = torch.device("cuda", local_rank)

    def to_dev(a):
        return torch.from_numpy(np.frombuffer(a.tobytes(), dtype=np.uint8).copy()).to(dev)

    d_rays = [to_dev(r) for r in ray_sets]
    d_poses = [to_dev(p) for p in pose_sets]
    d_queries = [to_dev(q) for q in query_sets]
    d_vfh_out = torch.zeros(nq * R.capi.VFH_OUT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    # output sets: the engine gives a batch to whichever stage has been free the longest, so a batch may still be running
    # when the pipeline has turned once more -- three turns' worth of buffers before one is written again
    n_out = 3 * args.pipeline
    d_paths = [torch.zeros(nq * args.max_path, dtype=torch.int32, device=dev) for _ in range(n_out)]
    d_results = [torch.zeros(nq * 6, dtype=torch.int32, device=dev) for _ in range(n_out)]
    e.vfh_init(nq)
    e.astar_pipeline_depth(args.pipeline)
    e.astar_configure(max_queries=nq, queue_capacity=args.queue_capacity, bucket_width=args.bucket_width)
    torch.cuda.synchronize()

    only_astar = os.environ.get("RNA_BENCH_ONLY_ASTAR") == "1"
    host_t = [0.0, 0.0, 0.0, 0] if os.environ.get("RNA_BENCH_HOST_TIMING") == "1" and not args.tiled else None
    step_no = [0]
    xfer = [0, 0]          # bytes received: halo strips, gathered windows

    def one_pass():
        b = step_no[0] % n_out
        k = step_no[0] % ROTATE
        step_no[0] += 1
        if only_astar:   # developer switch RNA_BENCH_ONLY_ASTAR=1: the search capacity without the map update and VFH+ (NOT the metric)
            e.astar_device(d_queries[k].data_ptr(), nq, d_paths[b].data_ptr(), args.max_path, d_results[b].data_ptr())
            return b
        if host_t is not None:   # developer switch RNA_BENCH_HOST_TIMING=1: host seconds inside each of the pass's three calls
            t_a = time.perf_counter()
            e.update_map_device(d_rays[k].data_ptr(), len(ray_sets[k]), compose_mode=0)
            t_b = time.perf_counter()
            e.vfh_step_device(d_poses[k].data_ptr(), nq, d_vfh_out.data_ptr())
            t_c = time.perf_counter()
            e.astar_device(d_queries[k].data_ptr(), nq, d_paths[b].data_ptr(), args.max_path, d_results[b].data_ptr())
            t_d = time.perf_counter()
            host_t[0] += t_b - t_a; host_t[1] += t_c - t_b; host_t[2] += t_d - t_c; host_t[3] += 1
            return b
        e.update_map_device(d_rays[k].data_ptr(), len(ray_sets[k]), compose_mode=0)
        if layout is not None and world > 1:
            xfer[0] += D.exchange_halo(e, R.capi.LAYER_MASTER, layout, rank, halo, dist, tracked=not args.tiled_full_gather)
        e.vfh_step_device(d_poses[k].data_ptr(), nq, d_vfh_out.data_ptr())
        if layout is not None and world > 1:
            if args.tiled_full_gather:
                xfer[1] += D.gather_layer(e, R.capi.LAYER_MASTER, layout, rank, dist)
            else:   # only the 64 x 64 tiles this update changed travel; their neighbour masks are refreshed, not all
                xfer[1] += D.gather_dirty(e, (R.capi.LAYER_LASER, R.capi.LAYER_MASTER), layout, rank, dist)
                e.compose_master(0)
        e.astar_device(d_queries[k].data_ptr(), nq, d_paths[b].data_ptr(), args.max_path, d_results[b].data_ptr())
        return b

    def step():   # one turn of the pipeline: every stage receives one batch
        for _ in range(args.pipeline):
            one_pass()

    def barrier():
        if use_dist:
            dist.barrier()

    def check_results(what):
        """every result buffer that may hold a batch: all queries answered (0 found / 1 no path), nothing else"""
        found = answered = total = 0
        for buf in d_results[:min(n_out, step_no[0])]:
            st = buf.cpu().numpy().reshape(nq, 6)[:, 0]
            bad = sorted(set(st[(st != 0) & (st != 1)].tolist()))
            if bad:
                raise SystemExit("A* batch did not complete %s: statuses %s (3 = path buffer too small, 4 = cost beyond 24 bits, "
                                 "5 = search pages used up, < 0 = frontier queue overflow)" % (what, bad))
            found += int((st == 0).sum())
            answered += int(((st == 0) | (st == 1)).sum())
            total += nq
        return found, answered, total

    for _ in range(args.warmup):
        step()
    e.synchronize()
    torch.cuda.synchronize()
    check_results("during warm-up")

    # HIP events bracket the dominant kernel (the A* search, on its stage's own stream) in the timed region.  The other
    # kernels all run on the engine stream, whose chain of short dependent kernels gates the next batch: an event record
    # is a packet of its own there (about 35 us each in the loop, ~1 ms per pass for the ten slots), so they are
    # bracketed in one extra turn of the pipeline AFTER the timed region instead (`kernel_ms_per_pass`, `roofline_rows`).
    e.profile(2)
    e.profile_reset()
    xfer[0] = xfer[1] = 0
    e.astar_job_counters(reset=True)   # (waits for the warm-up's searches; the timed region's jobs are read after it)
    if host_t is not None:
        host_t[:] = [0.0, 0.0, 0.0, 0]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    for _ in range(args.steps):
        step()
    e.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    host_cores_used = (time.process_time() - cpu0) / max(elapsed, 1e-9)   # CPU seconds of ALL the process's threads per wall second
    barrier()
    prof = e.profile_get()
    e.profile(False)
    job_counters = e.astar_job_counters(reset=True)
    host_t_timed = list(host_t) if host_t is not None else None
    found, answered, total = check_results("in the timed region")
    xfer_timed = list(xfer)
    e.profile(1)
    e.profile_reset()
    step()
    e.synchronize()
    torch.cuda.synchronize()
    prof_all = e.profile_get()
    e.profile(False)
    check_results("in the profiled turn after the timed region")
    xfer[0], xfer[1] = xfer_timed
    paths_checked = None
    if (args.check_paths or (world == 1 and not args.no_check_paths)) and layout is None:
        # the profiled turn's last pass: no map update has followed it, so the master layer is the map its snapshot saw
        b_last, k_last = (step_no[0] - 1) % n_out, (step_no[0] - 1) % ROTATE
        paths_checked = check_paths(e.download(R.capi.LAYER_MASTER), query_sets[k_last],
                                    d_results[b_last].cpu().numpy().reshape(nq, 6),
                                    d_paths[b_last].cpu().numpy().reshape(nq, args.max_path), n, n)
        if paths_checked["matched"] != paths_checked["queries"]:
            raise SystemExit("--check-paths: %d of %d queries of the checked batch differ from the oracle"
                             % (paths_checked["queries"] - paths_checked["matched"], paths_checked["queries"]))

    # E per query set (settled cells, from the pages still resident after a launch), on the map as the timed region left it
    settled_sets = []
    for k in range(ROTATE):
        e.astar_device(d_queries[k].data_ptr(), nq, d_paths[0].data_ptr(), args.max_path, d_results[0].data_ptr())
        e.synchronize()
        settled_sets.append(float(e.astar_settled(nq).astype(np.int64).sum()))
    settled_per_launch = float(np.mean(settled_sets))

    t_max = D.max_over_ranks(elapsed, "cpu" if share else dev) if use_dist else elapsed
    # every rank's shard of the global batch and the cycles it served in the timed region, for the line rank 0 prints
    shard_rows = [[lo, hi, nq * args.steps * args.pipeline]]
    windows = [list(layout.window(rank))] if layout is not None else None
    if use_dist and world > 1:
        t = torch.zeros(world, 7, dtype=torch.int64, device="cpu" if share else dev)
        t[rank, :3] = torch.tensor(shard_rows[0], dtype=torch.int64)
        if layout is not None:
            t[rank, 3:] = torch.tensor(windows[0], dtype=torch.int64)
        dist.all_reduce(t)
        shard_rows = t[:, :3].cpu().tolist()
        windows = t[:, 3:].cpu().tolist() if layout is not None else None

    if rank == 0:
        passes = args.steps * args.pipeline
        cycles = sum(r[2] for r in shard_rows)      # all ranks' cycles (= queries x world x passes)
        assert cycles == args.queries * world * passes
        ms_search = prof["astar_search"][0] / max(1, prof["astar_search"][1])
        alg_bytes = settled_per_launch * ASTAR_BYTES_PER_SETTLED
        achieved = alg_bytes / (ms_search * 1e-3) / 1e9 if ms_search > 0 else 0.0
        traffic, traffic_src = pmc_traffic("rna::tsa_search_kernel", args, world)
        wall_per_pass = t_max / passes

        def row(kernel_key, pmc_name, alg):
            ms = prof_all[kernel_key][0] / max(1, prof_all[kernel_key][1])
            gbs = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tr, src = pmc_traffic(pmc_name, args, world)
            return {"kernel": pmc_name if isinstance(pmc_name, str) else " + ".join(pmc_name), "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms,
                    "launches": prof_all[kernel_key][1], "measured": "one turn of the pipeline after the timed region"}

        himm_alg = float(np.mean([(8.0 * np.hypot(r["ex"] - r["sx"], r["ey"] - r["sy"]) / 0.05 + 8.0 * (r["clear_end"] == 0) + 40.0).sum()
                                  for r in ray_sets]))
        out = {
            "metric": ("replan cycles/sec (HIMM+VFH++A*) on %dx%d grid" if not only_astar else "DEVELOPER RUN, A* only, not the metric (%dx%d grid)") % (n, n),
            "value": cycles / t_max, "unit": "replan cycles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "full replan loop on %dx%d f32 grid (res 0.05 m); one step = one turn of the %d-stage A* "
                                   "pipeline = %d passes of [%d-ray HIMM batch + fused compose -> %d VFH+ poses -> %d grid-A* "
                                   "queries (one launch)]; 30%% rectangle obstacles (seed 2); %d ray batches / pose sets / "
                                   "query sets in rotation"
                                   % (n, n, args.pipeline, args.pipeline, len(ray_sets[0]), nq, nq, ROTATE),
                       "cycles_per_step": nq * args.pipeline, "passes_per_step": args.pipeline, "cycles_per_pass": nq,
                       "rays_per_pass": int(len(ray_sets[0])), "rotating_input_sets": ROTATE, "ms_per_pass": 1e3 * wall_per_pass,
                       "astar_queries_checked": total, "astar_queries_answered": answered, "astar_paths_found": found,
                       "astar_bucket_width": args.bucket_width or 128000, "astar_pipeline_depth": args.pipeline,
                       "astar_allocated": dict(zip(("pipeline_depth", "pages_per_query", "max_queries"), e.astar_effective_config())),
                       "timed_seconds": t_max, "paths_checked": paths_checked, "host_affinity": host_affinity,
                       # what a rank asks of the host in the timed region (eight ranks share one at N = 8): cores' worth of CPU time
                       "host_cores_used": host_cores_used,
                       "host_us_per_pass_developer": ({"update_map": 1e6 * host_t_timed[0] / host_t_timed[3], "vfh_step": 1e6 * host_t_timed[1] / host_t_timed[3],
                                                       "astar_batch (incl. the wait for a free stage)": 1e6 * host_t_timed[2] / host_t_timed[3]}
                                                      if host_t_timed and host_t_timed[3] else None),
                       "shards": [r[:2] for r in shard_rows], "cycles_by_rank": [r[2] for r in shard_rows],
                       "launcher": ({"spawned_by_bench": True, "parent_pid": int(os.environ["RNA_BENCH_PARENT"].split(":")[0]),
                                     "parent_is_my_parent": int(os.environ["RNA_BENCH_PARENT"].split(":")[0]) == os.getppid(),
                                     "parent_hip_free": os.environ["RNA_BENCH_PARENT"].split(":")[1] == "1"}
                                    if "RNA_BENCH_PARENT" in os.environ else {"spawned_by_bench": False}),
                       "parallelism": ("query-sharded x%d" % world) if layout is None else
                                      ("one map tiled %d x %d (windowed HIMM, %d-cell halo exchange, all-gather of owner "
                                       "windows), A* query-sharded x%d" % (layout.ti, layout.tj, halo, world))},
            "roofline": {"bound": "hbm", "kernel": "tsa_search_kernel (astar_search)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_provenance": profile_provenance(PMC_SUMMARY),
                         "algorithmic_bytes_per_launch": alg_bytes, "settled_cells_per_launch": settled_per_launch,
                         "settled_cells_by_query_set": settled_sets,
                         "avg_launch_ms": ms_search, "launches": prof["astar_search"][1],
                         # pipelined launches overlap on the GPU: each one's duration is stretched by the others, so
                         # the whole-step figure (algorithmic bytes of one launch / wall time of one step) is given too
                         "overlapped_launches": ms_search / (1e3 * wall_per_pass),
                         "achieved_per_pass_wall": alg_bytes / wall_per_pass / 1e9,
                         "frac_wall": alg_bytes / wall_per_pass / 1e9 / HBM_PEAK_GBS,
                         # the kernel's real bound (not HBM): see valu_issue()
                         "valu_issue": valu_issue(args, world, wall_per_pass),
                         "work_inflation": work_inflation(args, world, settled_per_launch, job_counters, passes)},
            "roofline_rows": [row("himm_raster", HIMM_RASTER_CHAIN, himm_alg),
                              row("vfh_step", "vfh_step_kernel", float(nq * VFH_BYTES_PER_POSE))],
            # every kernel slot bracketed by events, in ONE turn of the pipeline run after the timed region (the brackets
            # slow the engine stream down: these passes are slower than the timed ones)
            "kernel_ms_per_pass": {k: (v[0] / args.pipeline) for k, v in prof_all.items() if v[1]},
            "kernel_ms_per_pass_timed_region": {k: (v[0] / passes) for k, v in prof.items() if v[1]},
        }
        if layout is not None:
            out["tiled"] = {"layout": [layout.ti, layout.tj], "halo_cells": halo, "windows": windows,
                            "halo_bytes_per_pass_rank0": xfer[0] / passes,
                            "gather_bytes_per_pass_rank0": xfer[1] / passes}
        if not args.no_cpu and world == 1:   # rank 0 at N=1 only
            out["cpu_baseline"] = cb = cpu_baseline(args, R, master, ray_sets[0], pose_sets[0], query_sets[0], n, n, length)
            # the numerator both ways: E of the contract (plateau settled) and of an A* that stops at the goal (sample ratio)
            ratio = cb["settled_cells_stop_at_goal"] / max(1.0, cb["settled_cells_contract"])
            out["roofline"]["settled_ratio_stop_at_goal"] = ratio
            out["roofline"]["frac_wall_stop_at_goal_basis"] = out["roofline"]["frac_wall"] * ratio
        else:
            out["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    e.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
