"""The schedule of the grid-A* tile kernel (astar_tile.hip) as a CPU model, scripts/sim_dense2.c: 64 x 16-cell tiles,
rounds of tile jobs that read their halo as it was when the round began, alternating down / up sweeps with per-row
dirty flags and extra passes of a changed row along itself, f-buckets, rounds alternating between the two checkerboard
colours, and a neighbour woken only when an edge cell beats -- by a step its mask allows -- what the neighbour held.
The model checks itself against the oracle's cost and settled count E for every query; here it runs on maps small
enough for the CPU suite, with every combination of the options, so the exactness argument of DESIGN.md 5 is executed
without a GPU."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as O
from ros_navigation_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = "/tmp/rna_sim_dense2"
SIM_ASYNC = "/tmp/rna_sim_async"


@pytest.fixture(scope="module")
def sim():
    subprocess.check_call(["gcc", "-O2", "-o", SIM, os.path.join(ROOT, "scripts", "sim_dense2.c")])
    return SIM


def workload(path, rows, cols, nq, seed, density, side):
    master = synth.obstacles_rect(rows, cols, density=density, seed=seed, side=side)
    q = synth.astar_queries(nq, master, rows, cols, seed=seed)
    _, nbr = O.astar_masks(master, rows, cols)
    rec = np.zeros((nq, 4), np.int32)
    gw = np.empty(rows * cols, np.int32)
    for k in range(nq):
        res, _, _ = O.astar_query(nbr, rows, cols, q["start"][k], q["goal"][k], g_work=gw)
        rec[k] = (q["start"][k], q["goal"][k], res.cost, res.settled)
    with open(path, "wb") as f:
        np.array([rows, cols, nq], np.int32).tofile(f)
        nbr.tofile(f)
        rec.tofile(f)
    return rec


@pytest.mark.parametrize("rows,cols,density,side", [(200, 150, 0.30, (2, 20)), (333, 97, 0.45, (1, 6)), (130, 260, 0.15, (4, 40))])
def test_tile_schedule_model_matches_the_oracle(sim, tmp_path, rows, cols, density, side):
    wl = str(tmp_path / "wl.bin")
    rec = workload(wl, rows, cols, 24, seed=rows + cols, density=density, side=side)
    assert (rec[:, 2] < 0x7fffffff).sum() >= 12          # most queries have a path
    for bucket in (2828, 24000):
        for env in ({}, {"SIM_REDBLACK": "1"}, {"SIM_FILTER": "1"}, {"SIM_REDBLACK": "1", "SIM_FILTER": "1"}):
            for variant in ("0", "161"):
                out = subprocess.run([sim, wl, str(bucket), "24", variant], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
                assert out.returncode == 0 and "mismatches 0" in out.stdout, (bucket, env, variant, out.stdout[-400:], out.stderr[-400:])


@pytest.fixture(scope="module")
def sim_async():
    subprocess.check_call(["gcc", "-O2", "-o", SIM_ASYNC, os.path.join(ROOT, "scripts", "sim_async.c")])
    return SIM_ASYNC


@pytest.mark.parametrize("rows,cols,density,side", [(200, 150, 0.30, (2, 20)), (333, 97, 0.45, (1, 6))])
def test_open_list_schedule_model_matches_the_oracle(sim_async, tmp_path, rows, cols, density, side):
    """scripts/sim_async.c: the asynchronous schedule of tsa_search_kernel as a discrete-event model -- W wavefronts take
    the queued tile with the lowest key, a job reads the field as it is when it starts and publishes when it ends, a tile
    woken while it runs is queued again by its own wavefront (policy 4 is the kernel's protocol: pending / running bits,
    stale entries dropped; policies 1-3 are the alternatives it was chosen from, policy 0 the round-2 red-black rounds).
    Every policy, wavefront count and bucket width must reproduce the oracle's cost and settled count E exactly: the
    exactness argument (any order, as long as no wake-up is lost) executed without a GPU."""
    wl = str(tmp_path / "wl.bin")
    rec = workload(wl, rows, cols, 16, seed=rows + cols, density=density, side=side)
    assert (rec[:, 2] < 0x7fffffff).sum() >= 8
    for policy in ("0", "1", "2", "3", "4"):
        for bucket, waves in ((2828, 3), (24000, 8), (96000, 8), (400000, 16)):
            for env in ({}, {"SIM_KSHIFT": "2"}, {"SIM_DIRTYKEY": "2", "SIM_KSHIFT": "10"}) + (
                    ({"SIM_FIRSTROWS": "1", "SIM_KSHIFT": "9"}, {"SIM_FIRSTROWS": "1", "SIM_FIRSTWAKE": "1"}) if policy == "4" else ()):   # (round 5: the first-job rule of RNA_TSA_FIRST_ROWS)
                out = subprocess.run([sim_async, wl, str(bucket), "16", policy, str(waves)], capture_output=True, text=True,
                                     env=dict(os.environ, **env), timeout=300)
                assert out.returncode == 0 and "mismatches 0" in out.stdout, (policy, bucket, waves, env, out.stdout[-400:], out.stderr[-400:])
