"""Tiled single-map mode on CPU (SURVEY.md 8e mode 2): gloo ranks each own one window of a map, run the
halo exchange and the all-gather of ros_navigation_amd/dist.py on a numpy stand-in for the engine's
layers, and are checked against the oracle's HIMM update of the whole map."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS, COLS, HALO = 90, 70, 16          # not multiples of anything: windows of unequal size


class NumpyGrid:
    """pack_region / unpack_region of capi.Engine on a host array (column-major, linear = i + j*rows)."""

    def __init__(self, rows, cols, data):
        self.rows, self.cols = rows, cols
        self.a = data.reshape(cols, rows).copy()     # a[j, i]

    def pack_region(self, layer, i0, ni, j0, nj):
        import torch
        return torch.from_numpy(np.ascontiguousarray(self.a[j0:j0 + nj, i0:i0 + ni]).reshape(-1).copy())

    def unpack_region(self, layer, i0, ni, j0, nj, t):
        self.a[j0:j0 + nj, i0:i0 + ni] = t.numpy()[:ni * nj].reshape(nj, ni)


class NumpyTileGrid(NumpyGrid):
    """The tile-list calls of capi.Engine on host arrays: two layers (0 = laser, 1 = master), 64 x 64 tiles."""
    T = 64

    def __init__(self, rows, cols, data):
        super().__init__(rows, cols, data)
        self.layers = [self.a, self.a.copy()]
        self.flags = np.zeros(((cols + 63) // 64) * ((rows + 63) // 64), np.uint8)
        self.marked = set()

    def last_dirty_tiles(self):
        return self.flags.copy()

    def _clip(self, t, window):
        tiles_i = (self.rows + 63) // 64
        i0, ni, j0, nj = window
        a, b = (t % tiles_i) * 64, (t // tiles_i) * 64
        return max(a, i0), min(a + 64, i0 + ni, self.rows), max(b, j0), min(b + 64, j0 + nj, self.cols), a, b

    def pack_tiles(self, layer, tiles, window):
        import torch
        out = np.zeros((len(tiles), 64, 64), np.float32)          # [slot, lj, li]
        for k, t in enumerate(tiles):
            ia, ib, ja, jb, a, b = self._clip(int(t), window)
            if ia < ib and ja < jb:
                out[k, ja - b:jb - b, ia - a:ib - a] = self.layers[layer][ja:jb, ia:ib]
        return torch.from_numpy(out.reshape(-1))

    def unpack_tiles(self, layers, tiles, window, t):
        data = t.numpy().reshape(-1, 64, 64)
        for k, tl in enumerate(tiles):
            ia, ib, ja, jb, a, b = self._clip(int(tl), window)
            for l in layers:
                if ia < ib and ja < jb:
                    self.layers[l][ja:jb, ia:ib] = data[k, ja - b:jb - b, ia - a:ib - a]
            self.marked.add(int(tl))


def _dirty_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ros_navigation_amd import dist as D
    dist = D.init("gloo")
    rows, cols = 200, 150                                   # 4 x 3 tiles of 64, windows cut through tiles
    before, after = _scene(rows, cols)
    L = D.TileLayout.for_world(rows, cols, world)
    i0, ni, j0, nj = L.window(rank)
    grid = NumpyTileGrid(rows, cols, before)
    full, old = after.reshape(cols, rows), before.reshape(cols, rows)
    for lay in grid.layers:                                  # owner-computes, laser == master after compose
        lay[j0:j0 + nj, i0:i0 + ni] = full[j0:j0 + nj, i0:i0 + ni]
    ch = ~((full == old) | (np.isnan(full) & np.isnan(old)))
    win = np.zeros_like(ch)
    win[j0:j0 + nj, i0:i0 + ni] = True
    jj, ii = np.nonzero(ch & win)
    grid.flags[np.unique((jj // 64) * ((rows + 63) // 64) + ii // 64)] = 1     # what HIMM would have flagged
    got = D.gather_dirty(grid, (0, 1), L, rank, dist)
    ok = all(np.array_equal(lay, full, equal_nan=True) for lay in grid.layers)
    # tiles nobody changed were not touched, tiles a peer changed were flagged for the mask refresh
    peers = ch & ~win
    jj, ii = np.nonzero(peers)
    want_marked = set(np.unique((jj // 64) * ((rows + 63) // 64) + ii // 64).tolist())
    out.put((rank, ok, got, want_marked <= grid.marked, len(grid.marked), int(ch.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tiled_incremental_gather_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_dirty_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, got, marked_ok, n_marked, n_changed in res:
        assert ok, "rank %d: layers differ from the whole-map update after the incremental gather" % rank
        assert marked_ok and n_changed > 0
        assert got >= n_marked * 4096 * 4                                   # a tile two owners changed arrives twice
        assert got < 4 * 200 * 150 * 2                                      # well under two whole layers


def _scene(rows=ROWS, cols=COLS):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from ros_navigation_amd import synth
    g = O.make_geom(rows * 0.05, cols * 0.05, 0.05)
    before = synth.occupancy_sparse(rows, cols, seed=5)
    rays = synth.rays(6, 200, rows * 0.05, cols * 0.05, seed=9, lmin=0.3, lmax=3.0, margin=0.3)
    after = before.copy()
    O.himm_update(g, after, rays.view(O.RAY_DTYPE))
    return before, after


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ros_navigation_amd import dist as D
    dist = D.init("gloo")
    before, after = _scene()
    L = D.TileLayout.for_world(ROWS, COLS, world)
    i0, ni, j0, nj = L.window(rank)
    # owner-computes: this rank's layer holds the updated cells only inside its own window
    grid = NumpyGrid(ROWS, COLS, before)
    full = after.reshape(COLS, ROWS)
    grid.a[j0:j0 + nj, i0:i0 + ni] = full[j0:j0 + nj, i0:i0 + ni]
    got_halo = D.exchange_halo(grid, 0, L, rank, HALO, dist)
    fi0, fi1 = max(i0 - HALO, 0), min(i0 + ni + HALO, ROWS)
    fj0, fj1 = max(j0 - HALO, 0), min(j0 + nj + HALO, COLS)
    frame_ok = np.array_equal(grid.a[fj0:fj1, fi0:fi1], full[fj0:fj1, fi0:fi1], equal_nan=True)
    # outside the frame nothing may have been written
    mask = np.ones((COLS, ROWS), bool)
    mask[fj0:fj1, fi0:fi1] = False
    outside_ok = np.array_equal(grid.a[mask], before.reshape(COLS, ROWS)[mask], equal_nan=True)
    got_all = D.gather_layer(grid, 0, L, rank, dist)
    all_ok = np.array_equal(grid.a, full, equal_nan=True)
    out.put((rank, frame_ok, outside_ok, all_ok, got_halo, got_all, (fi1 - fi0) * (fj1 - fj0) - ni * nj))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tiled_halo_and_gather_gloo(world):
    before, after = _scene()
    assert not np.array_equal(before, after, equal_nan=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, frame_ok, outside_ok, all_ok, got_halo, got_all, frame_cells in res:
        assert frame_ok, "rank %d: window + halo frame differs from the whole-map update" % rank
        assert outside_ok, "rank %d: halo exchange wrote outside its frame" % rank
        assert all_ok, "rank %d: gathered layer differs from the whole-map update" % rank
        assert got_halo == 4 * frame_cells            # every frame cell received exactly once
    assert sum(r[5] for r in res) == 4 * ROWS * COLS * (world - 1)


def test_tile_layout_partition_and_owner():
    sys.path.insert(0, ROOT)
    from ros_navigation_amd.dist import TileLayout, vfh_halo
    assert vfh_halo(0.05) == 16                        # ceil(0.75 m / res) + 1
    for rows, cols, world in ((8192, 8192, 8), (90, 70, 4), (33, 47, 6), (10, 10, 1), (64, 64, 3)):
        L = TileLayout.for_world(rows, cols, world)
        assert L.world == world
        seen = np.full((rows, cols), -1)
        for r in range(world):
            i0, ni, j0, nj = L.window(r)
            assert (seen[i0:i0 + ni, j0:j0 + nj] == -1).all()
            seen[i0:i0 + ni, j0:j0 + nj] = r
        assert (seen >= 0).all()
        ii, jj = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
        assert np.array_equal(L.owner(ii, jj), seen)
    L = TileLayout.for_world(8192, 8192, 8)
    assert (L.ti, L.tj) == (2, 4) and L.window(5) == (4096, 4096, 2048, 2048)   # tile 4096 rows x 2048 cols
    with pytest.raises(ValueError):
        TileLayout(4, 4, 8, 1)
