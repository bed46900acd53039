"""bench.py prints ONE JSON line with the fields the driver reads (small grid so that it runs in seconds)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# one step = one turn of the pipeline: 3 stages -> 3 passes of [HIMM batch, 32 VFH+ poses, 32 A* queries] per step
SMALL = ["--grid", "512", "--queries", "32", "--steps", "5", "--warmup", "2", "--pipeline", "3", "--ray-poses", "8", "--rays-per-pose", "200"]
PASSES = 5 * 3


def run_bench(extra, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-1500:]     # ONE line on stdout (RCCL's version banner and the like go to stderr)
    return json.loads(lines[0])


def test_bench_line_contract_on_a_small_grid_through_the_spawn_path_with_rccl():
    """The JSON line's contract -- on the exact code an N-GPU run takes: a parent that never touches a GPU (GPUs counted
    from the KFD topology, no torch import) starts the rank with Popen, the rank initialises RCCL (`nccl` backend), runs
    the barriers around the timed region and the max over ranks -- forced at world size 1, un-shared, so that it has run
    on hardware (one subprocess serves both checks: a bench start-up costs a torch import and a map set-up each)."""
    d = run_bench(["--cpu-seconds", "0.5"], env={"RNA_BENCH_FORCE_SPAWN": "1"})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_rows", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 32 * PASSES / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    assert d["config"]["cycles_per_step"] == 32 * 3 and d["config"]["passes_per_step"] == 3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["achieved"] > 0 and r["launches"] == PASSES
    assert abs(r["frac_wall"] - r["achieved_per_pass_wall"] / r["peak"]) < 1e-12 and r["frac_wall"] > 0
    assert r["traffic"] is None      # the committed PMC passes belong to the default configuration, not to this one
    rows = {x["kernel"]: x for x in d["roofline_rows"]}
    assert len(rows) == 2 and "vfh_step_kernel" in rows and any("himm_tile_raster_kernel" in k for k in rows)
    for x in rows.values():
        assert x["achieved"] > 0 and x["launches"] == 3 and abs(x["frac"] - x["achieved"] / 8000.0) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    cfg = d["config"]
    # every query of every batch still in a result buffer was answered (found / no path), none failed
    assert cfg["astar_queries_checked"] == 32 * 3 * cfg["astar_pipeline_depth"] and cfg["astar_queries_answered"] == cfg["astar_queries_checked"]
    assert cfg["astar_paths_found"] > 0 and cfg["rotating_input_sets"] == 4
    # what the engine really allocated (stages / pages per query / concurrent queries may shrink to fit HBM) is reported
    alloc = cfg["astar_allocated"]
    assert alloc["pipeline_depth"] == cfg["astar_pipeline_depth"] and alloc["pages_per_query"] > 0 and alloc["max_queries"] == 32
    la = cfg["launcher"]
    assert la["spawned_by_bench"] and la["parent_is_my_parent"] and la["parent_hip_free"]
    assert cfg["shards"] == [[0, 32]] and cfg["cycles_by_rank"] == [32 * PASSES]


# The 8-GPU bench runs once, unattended, at the end of a round: its two launch modes are rehearsed here at world size 8
# with all ranks sharing the test box's GPU (RCCL refuses two ranks on one device, so the barrier and the max over ranks
# go through gloo -- the RCCL leg of the same code runs in the test above), on a 1024 x 1024 grid.
WORLD8 = ["--grid", "1024", "--queries", "32", "--steps", "3", "--warmup", "1", "--ray-poses", "8", "--rays-per-pose", "200",
          "--gpus", "8", "--no-cpu"]


def run_world8(extra, pipeline=3):
    extra = extra + (["--pipeline", str(pipeline)] if pipeline else [])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + WORLD8 + extra, capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=dict(os.environ, RNA_BENCH_SHARE_GPU="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]      # ONE line, from rank 0, nothing else on stdout
    return json.loads(lines[0])


def check_world8_common(d, depth=3):
    cfg = d["config"]
    passes = 3 * depth
    assert d["n_gpus"] == 8 and d["cpu_baseline"] is None and d["scaling"] == "weak"
    la = cfg["launcher"]
    assert la["spawned_by_bench"] and la["parent_is_my_parent"] and la["parent_hip_free"]      # 8 ranks, started by a HIP-free parent
    # the shards of the global batch (8 x 32 cycles per pass): disjoint, complete, in rank order
    assert cfg["shards"] == [[32 * r, 32 * (r + 1)] for r in range(8)]
    assert cfg["cycles_by_rank"] == [32 * passes] * 8
    # value = the cycles ALL ranks served / the slowest rank's time
    total = sum(cfg["cycles_by_rank"])
    assert abs(d["value"] - total / cfg["timed_seconds"]) < 1e-6 * d["value"]
    assert abs(d["ms_per_step"] * 3e-3 - cfg["timed_seconds"]) < 1e-9
    assert cfg["astar_queries_answered"] == cfg["astar_queries_checked"] and cfg["astar_paths_found"] > 0
    assert cfg["astar_pipeline_depth"] == depth == cfg["astar_allocated"]["pipeline_depth"] and cfg["passes_per_step"] == depth
    # every rank confined itself to host cores before it loaded torch / HIP (the GPU's NUMA node where the topology names one)
    assert cfg["host_affinity"]["cpus"] >= 1 and ("error" not in cfg["host_affinity"] or not cfg["host_affinity"]["pinned"]), cfg["host_affinity"]


def test_world8_rehearsal_query_sharded():
    """SURVEY 8e mode 1 as the driver launches it at N = 8 (`bench.py --gpus 8`, NO --pipeline): replicated maps, sharded
    cycles, and the pipeline depth the bench takes by itself next to a communicator (DEFAULT_PIPELINE_RCCL, not the one-GPU
    default) -- the configuration of the driver's run is the one rehearsed"""
    import bench
    d = run_world8([], pipeline=0)
    check_world8_common(d, depth=bench.DEFAULT_PIPELINE_RCCL)
    assert bench.DEFAULT_PIPELINE_RCCL != bench.DEFAULT_PIPELINE
    assert "query-sharded x8" in d["config"]["parallelism"] and "tiled" not in d


def test_world8_rehearsal_one_map_tiled_2x4():
    """SURVEY 8e mode 2 / BASELINE config 5's layout (`--gpus 8 --tiled`): one map in 2 x 4 windows, every window at
    least as wide as the VFH+ halo, the windows a partition of the map, strips and dirty tiles really travelling"""
    d = run_world8(["--tiled"])
    check_world8_common(d)
    t = d["tiled"]
    assert t["layout"] == [2, 4] and t["halo_cells"] == 16
    wins = t["windows"]
    assert len(wins) == 8 and all(w[1] >= t["halo_cells"] and w[3] >= t["halo_cells"] for w in wins)
    cover = np.zeros((1024, 1024), np.int32)
    for i0, ni, j0, nj in wins:
        cover[i0:i0 + ni, j0:j0 + nj] += 1
    assert (cover == 1).all()
    assert wins == [[512 * (r // 4), 512, 256 * (r % 4), 256] for r in range(8)]
    assert t["halo_bytes_per_pass_rank0"] > 0 and t["gather_bytes_per_pass_rank0"] > 0
    assert "one map tiled 2 x 4" in d["config"]["parallelism"]


def test_parent_of_the_ranks_counts_gpus_without_opening_hip():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    n = bench.visible_gpus()
    assert n is None or n >= 1
    env = dict(os.environ, ROCR_VISIBLE_DEVICES="0")
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print(bench.visible_gpus()); "
                          "print('torch' in sys.modules, 'ros_navigation_amd' in sys.modules)" % ROOT],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    first, second = out.stdout.split("\n")[:2]
    assert first.strip() in ("1", "None") and second.strip() == "False False"


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--gpus", "2"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


_default_runs = {}


def default_bench(cpus=0):
    """the default bench (full size, 12 steps, --check-paths), run once per setting and shared by the tests below"""
    if cpus not in _default_runs:
        env = dict(os.environ, **({"RNA_BENCH_CPUS": str(cpus)} if cpus else {}))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--no-cpu"] + ([] if cpus else ["--check-paths"]),
                             capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        _default_runs[cpus] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    return _default_runs[cpus]


def test_default_bench_holds_its_rate_on_four_host_cores():
    """Eight ranks share one host at N = 8 (SURVEY 8e mode 1): a rank must not need more than its share of the cores.  The
    default bench with the process confined to FOUR host cores (RNA_BENCH_CPUS=4: main thread, the engine's launch
    thread, the runtime's helpers) keeps >= 95 % of the unconfined rate on the same box."""
    # (some sandboxes refuse sched_setaffinity: the bench then reports the refusal in config.host_affinity and runs unpinned --
    # nothing to compare; asked of the bench's own function in a fresh process, exactly as a rank would call it)
    probe = subprocess.run([sys.executable, "-c", "import json, bench; print(json.dumps(bench.pin_to_gpu_numa_node(0)))"],
                           capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, RNA_BENCH_CPUS="4"))
    info = json.loads(probe.stdout.strip().splitlines()[-1]) if probe.returncode == 0 and probe.stdout.strip() else {"error": probe.stderr[-300:]}
    if info.get("cpus") != 4:
        pytest.skip("this box does not let a process confine itself to four cores: %r" % (info,))
    free, four = default_bench(), default_bench(cpus=4)
    assert four["config"]["host_affinity"]["cpus"] == 4, four["config"]["host_affinity"]
    assert four["value"] >= 0.95 * free["value"], (four["value"], free["value"])


def test_default_bench_keeps_the_engine_stream_alive():
    """The full-size loop, shortened: the search streams' CU mask has to leave the short engine-stream kernels (map
    update, VFH+, field reset) somewhere to run -- when four search workgroups per CU took every register of every CU,
    himm_prep took 3-6 ms instead of 0.3 ms per step and the step rate hung on the engine stream (22 k cycles/s).  Loose
    bounds: a guard against that cliff and against the hardware-queue cliff of too many pipeline stages, not a benchmark."""
    d = default_bench()
    k = d["kernel_ms_per_pass"]
    assert k["himm_prep"] < 1.0 and k["vfh_step"] < 0.6 and k["compose_master"] < 0.6, k
    # the engine stream's chain (astar_search / astar_reset / astar_init run on the stages' own streams, vfh_step on the VFH+ stream)
    engine = sum(v for name, v in k.items() if name not in ("astar_search", "astar_reset", "astar_init", "vfh_step"))
    # (measured 1.08 - 1.22 ms in round 4 -- 1.33 - 1.43 in round 3 --, bracketed by events that cost ~35 us per kernel
    # themselves; the same kernels take 0.5 ms on the 32 reserved CUs alone, the rest is waiting for wave slots)
    assert engine < 1.5 and engine < d["config"]["ms_per_pass"], (k, d["config"]["ms_per_pass"])
    # a cliff guard, not a benchmark: the rate when the engine stream starved was 22 k, with too few hardware queues 31 k
    # (this kernel runs at 125 k+); RNA_TEST_BENCH_FLOOR overrides it on a shared or throttled box
    assert d["value"] > float(os.environ.get("RNA_TEST_BENCH_FLOOR", "50000")), d["value"]
    assert d["config"]["astar_allocated"]["pipeline_depth"] == d["config"]["astar_pipeline_depth"]
    # parity AT the headline configuration (--check-paths): one whole batch of the loop -- searched next to the other batches of
    # a full turn of the default pipeline -- path for path against the oracle
    import bench
    assert d["config"]["astar_pipeline_depth"] == bench.DEFAULT_PIPELINE
    pc = d["config"]["paths_checked"]
    assert pc["queries"] == 256 and pc["matched"] == 256 and pc["paths_found"] > 200, pc
    # eight ranks share one host at N = 8: what ONE rank asks of it in the timed region -- CPU seconds of all its threads (main
    # thread, the engine's launch thread, the runtime's helpers) per wall second -- has to stay well below an eighth of any host
    # the driver would use (independent of whether the sandbox lets a process pin itself, see the four-core test above)
    assert 0.0 < d["config"]["host_cores_used"] < 4.0, d["config"]["host_cores_used"]


def test_bench_under_torchrun_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` -- the launcher of the driver's N > 1 runs
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, no spawning by bench.py itself) -- with the two ranks
    sharing the test box's GPU: one JSON line on stdout, both ranks' cycles in `value`."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", os.path.join(ROOT, "bench.py")] + SMALL + ["--gpus", "2", "--no-cpu"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, RNA_BENCH_SHARE_GPU="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["launcher"] == {"spawned_by_bench": False}
    assert d["config"]["shards"] == [[0, 32], [32, 64]] and d["config"]["cycles_by_rank"] == [32 * PASSES] * 2
    assert abs(d["value"] - 2 * 32 * PASSES / d["config"]["timed_seconds"]) < 1e-6 * d["value"]
