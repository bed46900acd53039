"""Multi-GPU plumbing, one process per GPU.
Mode 1 (default): queries/poses sharded across ranks with NO data-path collective (the grid is
replicated; every rank applies the same ray batch); torch.distributed only carries the benchmark
barrier and the max-over-ranks timing, as the bench contract requires.
Mode 2 (opt-in, `bench.py --tiled`): one map tiled over the GPUs with a halo exchange and an
all-gather of the owner windows -- see TileLayout below."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) share of n independent queries for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if device is not None:
        dist.init_process_group(backend, device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def broadcast_rays(rays, src=0):
    """Mode 1 in deployment: the rank that owns the sensors hands the ray batch (a uint8 / structured tensor of
    rna_ray records, 40 B each -- 4 MB for 100 k rays) to every replica before they all apply it.  One
    broadcast (RCCL over xGMI under "nccl"); the synthetic bench does not need it (every rank derives the batch)."""
    import torch.distributed as dist
    dist.broadcast(rays, src=src)
    return rays


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (elapsed seconds)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ---------------------------------------------------------------------------------------------------
# Tiled single map (SURVEY.md 8e mode 2, BASELINE config 5): one map split into ti x tj windows, one
# per GPU.  Every GPU holds a full-size layer but is authoritative only for its window:
#   HIMM      every rank receives the whole ray batch; rna_himm_set_window makes it write only its own
#             cells of the globally rasterised lines -- no exchange, bit-identical inside the window;
#   VFH+      a pose is served by the rank that owns its cell; the 1.5 m submap reaches <= halo cells
#             into the neighbours, so after each map update the window edges travel to the (up to 8)
#             neighbours: exchange_halo, two rounds of send/recv (RCCL over xGMI under "nccl");
#   grid A*   searches cross the whole map: gather_layer all-gathers the owner windows into every
#             rank's master layer once per update, then the queries are sharded as in mode 1.
# `grid` is anything with pack_region(layer, i0, ni, j0, nj) -> dense float32 tensor (i fastest) and
# unpack_region(layer, i0, ni, j0, nj, tensor): capi.Engine on the GPU, a numpy stand-in in CPU tests.
# ---------------------------------------------------------------------------------------------------
class TileLayout:
    """rows x cols buffer-index space cut into ti x tj windows; rank = a * tj + b owns window (a, b)."""

    def __init__(self, rows, cols, ti, tj):
        if ti < 1 or tj < 1 or ti > rows or tj > cols:
            raise ValueError("TileLayout: %d x %d windows do not fit a %d x %d map" % (ti, tj, rows, cols))
        self.rows, self.cols, self.ti, self.tj = rows, cols, ti, tj

    @classmethod
    def for_world(cls, rows, cols, world):
        """BASELINE config 5's shape: 2 x world/2 (8 GPUs -> 2 x 4); odd world sizes -> 1 x world."""
        ti = 2 if world % 2 == 0 else 1
        return cls(rows, cols, ti, world // ti)

    @property
    def world(self):
        return self.ti * self.tj

    def coords(self, rank):
        return divmod(rank, self.tj)

    def rank_of(self, a, b):
        return a * self.tj + b if 0 <= a < self.ti and 0 <= b < self.tj else None

    def window(self, rank):
        """(i0, ni, j0, nj) of the window `rank` owns."""
        a, b = self.coords(rank)
        i0, i1 = shard_bounds(self.rows, a, self.ti)
        j0, j1 = shard_bounds(self.cols, b, self.tj)
        return i0, i1 - i0, j0, j1 - j0

    def owner(self, i, j):
        """Rank that owns cell (i, j) (numpy arrays accepted)."""
        import numpy as np
        ib = np.array([shard_bounds(self.rows, a, self.ti)[1] for a in range(self.ti)])
        jb = np.array([shard_bounds(self.cols, b, self.tj)[1] for b in range(self.tj)])
        return np.searchsorted(ib, i, side="right") * self.tj + np.searchsorted(jb, j, side="right")


def vfh_halo(resolution, submap_length=1.5):
    """Cells a Steerer submap (1.5 m, mc/src/steerer.cpp:128-135) can reach past its centre cell."""
    import math
    return int(math.ceil(0.5 * submap_length / resolution)) + 1


def _p2p(dist, sends, recvs):
    """sends / recvs: lists of (tensor, peer).  One batch of non-blocking send/recv; device tensors go
    straight to RCCL, and are staged through the host under gloo (CPU tests, one-GPU developer runs)."""
    import torch
    if not sends and not recvs:
        return
    stage = dist.get_backend() == "gloo"
    ops, back = [], []
    for t, peer in sends:
        ops.append(dist.P2POp(dist.isend, t.cpu() if (stage and t.is_cuda) else t, peer))
    for t, peer in recvs:
        if stage and t.is_cuda:
            h = torch.empty(t.shape, dtype=t.dtype)
            back.append((t, h))
            t = h
        ops.append(dist.P2POp(dist.irecv, t, peer))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for t, h in back:
        t.copy_(h)


def exchange_halo(grid, layer, layout, rank, halo, dist, tracked=False):
    """Bring the `halo`-cell frame around this rank's window up to date from its owners.  Round 1 moves
    row strips (i direction) of the own columns, round 2 column strips that include the rows just
    received, which carries the corners: 4 messages per rank instead of 8.  Returns the bytes received."""
    import torch
    a, b = layout.coords(rank)
    i0, ni, j0, nj = layout.window(rank)
    got = 0

    def one_round(strips):
        nonlocal got
        sends, recvs, into = [], [], []
        for peer, send_box, recv_box in strips:
            if peer is None:
                continue
            sends.append((grid.pack_region(layer, *send_box), peer))
            buf = torch.empty(recv_box[1] * recv_box[3], dtype=torch.float32, device=sends[-1][0].device)
            recvs.append((buf, peer))
            into.append((recv_box, buf))
        _p2p(dist, sends, recvs)
        for box, buf in into:
            if tracked:     # per-tile bookkeeping (incremental mode, see gather_dirty)
                grid.unpack_region(layer, box[0], box[1], box[2], box[3], buf, tracked=True)
            else:
                grid.unpack_region(layer, box[0], box[1], box[2], box[3], buf)
            got += buf.numel() * 4

    up, down = layout.rank_of(a - 1, b), layout.rank_of(a + 1, b)
    for peer in (up, down):
        if peer is not None and layout.window(peer)[1] < halo:
            raise ValueError("halo of %d cells is wider than a neighbouring window" % halo)
    if ni < halo or nj < halo:
        raise ValueError("halo of %d cells is wider than this rank's window" % halo)
    one_round([(up, (i0, halo, j0, nj), (i0 - halo, halo, j0, nj)),
               (down, (i0 + ni - halo, halo, j0, nj), (i0 + ni, halo, j0, nj))])
    ie0 = i0 - (halo if up is not None else 0)
    ie1 = i0 + ni + (halo if down is not None else 0)
    left, right = layout.rank_of(a, b - 1), layout.rank_of(a, b + 1)
    for peer in (left, right):
        if peer is not None and layout.window(peer)[3] < halo:
            raise ValueError("halo of %d cells is wider than a neighbouring window" % halo)
    one_round([(left, (ie0, ie1 - ie0, j0, halo), (ie0, ie1 - ie0, j0 - halo, halo)),
               (right, (ie0, ie1 - ie0, j0 + nj - halo, halo), (ie0, ie1 - ie0, j0 + nj, halo))])
    return got


def gather_layer(grid, layer, layout, rank, dist):
    """All-gather of the owner windows: afterwards `layer` is complete and identical on every rank.
    Windows are padded to the largest one (collectives want equal counts).  Returns the bytes received."""
    import torch
    if layout.world == 1:
        return 0
    wins = [layout.window(r) for r in range(layout.world)]
    cap = max(w[1] * w[3] for w in wins)
    i0, ni, j0, nj = wins[rank]
    mine = grid.pack_region(layer, i0, ni, j0, nj)
    if mine.numel() < cap:
        mine = torch.cat([mine, mine.new_zeros(cap - mine.numel())])
    stage = dist.get_backend() == "gloo" and mine.is_cuda
    src = mine.cpu() if stage else mine
    parts = [torch.empty_like(src) for _ in range(layout.world)]
    dist.all_gather(parts, src)
    got = 0
    for r, (ri0, rni, rj0, rnj) in enumerate(wins):
        if r == rank:
            continue
        part = parts[r][:rni * rnj]
        grid.unpack_region(layer, ri0, rni, rj0, rnj, part.to(mine.device).contiguous() if stage else part.contiguous())
        got += rni * rnj * 4
    return got


def _gather_dirty_device(grid, layers, layout, rank, dist, tile):
    """gather_dirty with the tile lists resident on the GPU (RCCL backend): the flags are compacted into a device list
    (rna_last_dirty_tiles_device), lists and tile data are all-gathered as device tensors and unpacked from device lists;
    only the `world` counters visit the host, to size the collectives -- the same sequence as rna_rccl_gather_dirty
    (csrc/rccl_tiled.hip) runs for a C++ host."""
    import torch
    dev = grid._torch_device()
    tiles_i, tiles_j = grid.tile_grid()
    win = layout.window(rank)
    lst_all = torch.empty(tiles_i * tiles_j, dtype=torch.int32, device=dev)
    # (torch.empty: nothing torch's stream would still have to write -- the engine memsets the counter on its own stream,
    # which nothing orders with a torch fill kernel)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    torch.cuda.current_stream(dev).synchronize()         # (the allocator may hand out memory a torch kernel is still using)
    grid.last_dirty_tiles_device(win, lst_all.data_ptr(), count.data_ptr())
    # the MAP stream wrote list and count; torch's stream reads them.  Only that stream is waited for: the A* batches
    # of earlier passes stay in flight (rna_synchronize would drain the whole search pipeline on every pass)
    grid.synchronize_map()
    counts_t = [torch.zeros_like(count) for _ in range(layout.world)]
    dist.all_gather(counts_t, count)
    counts = [int(c.item()) for c in counts_t]           # 4 * world bytes: the one host visit
    cap = max(counts)
    if cap == 0:
        return 0
    lst = lst_all[:cap].contiguous()
    lists = [torch.empty_like(lst) for _ in range(layout.world)]
    dist.all_gather(lists, lst)
    pad = torch.zeros(cap * tile * tile, dtype=torch.float32, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    grid.pack_tiles_device(layers[0], lst.data_ptr(), counts[rank], win, pad.data_ptr())
    grid.synchronize_map()
    parts = [torch.empty_like(pad) for _ in range(layout.world)]
    dist.all_gather(parts, pad)
    torch.cuda.current_stream(dev).synchronize()
    got = 0
    for r in range(layout.world):
        if r == rank or counts[r] == 0:
            continue
        grid.unpack_tiles_device(layers, lists[r].data_ptr(), counts[r], layout.window(r), parts[r].data_ptr())
        got += counts[r] * tile * tile * 4
    grid.synchronize_map()                               # lists / parts may be freed by torch after this call returns
    return got


def gather_dirty(grid, layers, layout, rank, dist, tile=64):
    """Incremental form of gather_layer: every rank hands on only the 64 x 64 tiles its last map update changed
    (grid.last_dirty_tiles(), clipped to its window), and the receivers write them into `layers` (laser and master, so
    that the laser layer stays complete everywhere) with per-tile bookkeeping -- a following grid.compose_master(0)
    refreshes the A* neighbour masks of exactly those tiles.  Three small collectives (counts, tile lists) and one
    all-gather of the tile data, padded to the largest contribution.  Returns the bytes received."""
    import numpy as np
    import torch
    if layout.world == 1:
        return 0
    if dist.get_backend() == "nccl":
        return _gather_dirty_device(grid, layers, layout, rank, dist, tile)
    tiles_i = (layout.rows + tile - 1) // tile
    flags = grid.last_dirty_tiles()
    win = layout.window(rank)
    t = np.flatnonzero(flags)
    ti, tj = t % tiles_i, t // tiles_i
    i0, ni, j0, nj = win
    mine = t[(ti * tile < i0 + ni) & (ti * tile + tile > i0) & (tj * tile < j0 + nj) & (tj * tile + tile > j0)].astype(np.int32)
    probe = grid.pack_tiles(layers[0], mine[:0], win)                   # an empty tensor on the grid's device
    stage = dist.get_backend() == "gloo" and probe.is_cuda
    dev = torch.device("cpu") if stage else probe.device
    count = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(layout.world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    cap = max(counts)
    if cap == 0:
        return 0
    lst = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    lst[:len(mine)] = torch.from_numpy(mine).to(dev)
    lists = [torch.empty_like(lst) for _ in range(layout.world)]
    dist.all_gather(lists, lst)
    data = grid.pack_tiles(layers[0], mine, win)
    pad = torch.zeros(cap * tile * tile, dtype=torch.float32, device=dev)
    pad[:data.numel()] = data.to(dev)
    parts = [torch.empty_like(pad) for _ in range(layout.world)]
    dist.all_gather(parts, pad)
    got = 0
    for r in range(layout.world):
        if r == rank or counts[r] == 0:
            continue
        tiles_r = lists[r][:counts[r]].cpu().numpy()
        part = parts[r][:counts[r] * tile * tile]
        grid.unpack_tiles(layers, tiles_r, layout.window(r), part.to(probe.device).contiguous() if stage else part.contiguous())
        got += counts[r] * tile * tile * 4
    return got
