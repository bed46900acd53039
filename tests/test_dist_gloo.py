"""N > 1 path on CPU: two gloo ranks shard a query batch, time a fake step and reduce exactly as
bench.py does (barrier, max-over-ranks time, whole-job cycle count)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ros_navigation_amd import dist as D, synth
    r, lr, w = D.env_rank_world()
    assert (r, lr, w) == (rank, rank, world)
    dist = D.init("gloo")
    # every rank derives the same global batch, then takes its shard (no collective on the data path)
    master = synth.obstacles_rect(96, 96, density=0.2, seed=2, side=(2, 8))
    q = synth.astar_queries(101, master, 96, 96, seed=2)
    lo, hi = D.shard_bounds(len(q), r, w)
    mine = q[lo:hi]
    # deployment path of mode 1: rank 0 owns the sensors and broadcasts the ray batch to the replicas
    import torch
    rays = synth.rays(3, 50, 4.8, 4.8, seed=4, lmin=0.2, lmax=2.0, margin=0.2)
    t = torch.from_numpy(np.frombuffer(rays.tobytes(), dtype=np.uint8).copy())
    if r != 0:
        t.zero_()
    D.broadcast_rays(t, src=0)
    assert t.numpy().tobytes() == rays.tobytes()
    dist.barrier()
    elapsed = 0.25 + 0.5 * rank            # rank 1 is the slow one
    t_max = D.max_over_ranks(elapsed)
    total = D.sum_over_ranks(len(mine))
    out.put((rank, lo, hi, t_max, total, int(mine["start"].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, n0, s0), (r1, lo1, hi1, t1, n1, s1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 51, 51, 101)          # contiguous, balanced, complete
    assert t0 == t1 == 0.75 and n0 == n1 == 101              # MAX over ranks, whole-job count
    sys.path.insert(0, ROOT)
    from ros_navigation_amd import synth
    master = synth.obstacles_rect(96, 96, density=0.2, seed=2, side=(2, 8))
    q = synth.astar_queries(101, master, 96, 96, seed=2)
    assert s0 + s1 == int(q["start"].sum())                  # shards are a partition of the batch


def test_shard_bounds_partition():
    from ros_navigation_amd.dist import shard_bounds
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
