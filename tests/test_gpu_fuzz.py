"""Randomised parity in the driver-run suite: the time-boxed fuzzers of scripts/ (the same code the developer runs for
hours) with a FIXED seed and a short budget each, as subprocesses -- random map shapes that are not multiples of the
64 x 16 search tiles, densities 0..0.6, unknown cells, bucket widths 2828..400000, chunked batches, engine reuse, maps
moved once or twice (grid A*); random geometry / resolution / moved buffers, end points on cell centres, edges and the
map border, poses past the border (map update + VFH+).  A fuzzer exits non-zero at the first difference from the oracle
and prints the configuration that reproduces it.  The search kernel's correctness rests on an asynchronous scheduler,
`asm volatile` fences and on what the optimiser may hoist: this net catches what the hand-picked cases do not."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_fuzzer(script, seconds, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(seconds), str(seed)], capture_output=True, text=True,
                         timeout=seconds + 240, cwd=ROOT)
    assert out.returncode == 0, (script, seed, out.stdout[-3000:], out.stderr[-2000:])
    assert "fuzz ok" in out.stdout, out.stdout[-1000:]
    return out.stdout


@pytest.mark.parametrize("seed", [301, 302])
def test_fuzz_grid_astar_fixed_seed(seed):
    text = run_fuzzer("fuzz_astar.py", 15, seed)
    maps = int(text.split("):")[1].split("maps")[0])
    assert maps >= 10, text     # the budget was spent on searches, not on start-up


def test_fuzz_map_update_and_vfh_fixed_seed():
    run_fuzzer("fuzz_himm_vfh.py", 15, 303)
