"""Developer tool: time-boxed random parity run of the window-restricted HIMM update (tiled single-map mode,
rna_himm_set_window) against the CPU oracle's whole-map update: random geometry (size, resolution, map position, moved
buffer), random tile layouts (1..9 windows, unequal sizes), ray batches with end points snapped to cell centres / edges /
the map border.  For every window: inside == the whole-map result, outside == untouched, bit for bit; the union of the
windows is therefore the whole-map update.
usage: python scripts/fuzz_tiled.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
import ros_navigation_amd as R  # noqa: E402
from ros_navigation_amd.dist import TileLayout  # noqa: E402
import _oracle as O  # noqa: E402
from fuzz_himm_vfh import gen_rays, same_f32  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = nrays = nwin = 0
    while time.time() < t_end:
        res = float(rng.choice([0.05, 0.05, 0.1, 0.2, 0.025]))
        lx, ly = float(rng.uniform(3, 20)), float(rng.uniform(3, 20))
        px, py = (float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))) if rng.random() < 0.5 else (0.0, 0.0)
        here = dict(res=res, lx=lx, ly=ly, px=px, py=py, fuzz_seed=seed, case=cases)
        e = R.Engine(lx, ly, res, px, py)
        g = O.make_geom(lx, ly, res, px, py)
        ref = rng.choice(np.array([np.nan, 0, 10, 50, 150, 160, 170, 180, 7.5, -3, 1e3], np.float32), e.ncell)
        for l in range(3):
            e.upload(l, ref)
        if rng.random() < 0.4:
            target = (px + float(rng.uniform(-3, 3)), py + float(rng.uniform(-3, 3)))
            ptrs = (C.POINTER(C.c_float) * 1)(O.fptr(ref))
            regs = (O.Region * 4)()
            mv = C.c_int(0)
            O.lib().og_move(C.byref(g), ptrs, 1, O.d2(*target), regs, C.byref(mv))
            e.move(*target)
            here["moved"] = target
        ti, tj = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        L = TileLayout(e.rows, e.cols, ti, tj)
        for b in range(int(rng.integers(1, 3))):
            n = int(rng.choice([1, 17, 400, 5000]))
            rays = gen_rays(rng, n, g, lx, ly, res)
            before = ref.copy()
            O.himm_update(g, ref, rays)
            B, A = before.reshape(e.cols, e.rows), ref.reshape(e.cols, e.rows)
            for rank in range(L.world):
                i0, ni, j0, nj = L.window(rank)
                e.upload(R.capi.LAYER_LASER, before)
                e.himm_set_window(i0, j0, ni, nj)
                e.himm_update(R.capi.LAYER_LASER, rays.view(R.capi.RAY_DTYPE))
                want = B.copy()
                want[j0:j0 + nj, i0:i0 + ni] = A[j0:j0 + nj, i0:i0 + ni]
                got = e.download(R.capi.LAYER_LASER).reshape(e.cols, e.rows)
                if not same_f32(got, want):
                    bad = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
                    print("MISMATCH windowed himm", here, "layout", (ti, tj), "rank", rank, "window", (i0, ni, j0, nj), "batch", b,
                          "n", n, "cells (j, i)", bad[:5].tolist(), "of", len(bad))
                    sys.exit(1)
                nwin += 1
            nrays += n
        cases += 1
        e.close()
    print("tiled himm fuzz ok: %d maps, %d windows, %d rays in %.0f s, seed %d" % (cases, nwin, nrays, budget, seed))


if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    main()
