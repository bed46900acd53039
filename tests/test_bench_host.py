"""CPU-only: the parts of bench.py that need no GPU -- what the result line says about the committed profile files it quotes,
and the work figures it derives from the search kernels' own job counters (rna_astar_job_counters)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_committed_profiles_describe_the_committed_search_kernel():
    """bench.py quotes HBM traffic (PMC passes) and instruction counts (SQ counter pass) from files under profiles/; each carries
    the sha256[:12] of ros_navigation_amd/csrc/astar_tile.hip it was taken at, and the line says `stale` when that is not the
    source in the tree.  The committed ones must not be stale: a kernel change goes with scripts/profile_r06.sh."""
    sha = bench.kernel_source_sha()
    for path in (bench.PMC_SUMMARY, bench.SQ_COUNTERS):
        p = bench.profile_provenance(path)
        assert p["kernel_source_sha"] == sha and p["stale"] is False, p
    d = json.load(open(bench.PMC_SUMMARY))
    assert d["config"]["pipeline"] == bench.DEFAULT_PIPELINE      # the passes were taken at the depth the bench runs by default


def test_work_inflation_from_job_counters():
    args = argparse.Namespace(grid=4096, queries=256, pipeline=bench.DEFAULT_PIPELINE, tiled=False)
    c = dict(searches=1000, tiles_touched=669000, jobs=2709000, jobs_noop=910000, sticky_turns=975000, rows_written=24000000,
             buckets=1300, bucket_reruns=11)
    w = bench.work_inflation(args, 1, 1.0e8, c, passes=1000 // 256 + 1)
    assert abs(w["jobs_per_touched_tile"] - 2709000 / 669000) < 1e-12 and abs(w["noop_job_frac"] - 910 / 2709) < 1e-12
    assert abs(w["sticky_turn_frac"] - 975 / 2709) < 1e-12 and abs(w["rows_written_per_job"] - 24000 / 2709) < 1e-9
    assert w["observed_in"].startswith("the timed region") and w["instructions_provenance"]["stale"] is False
    assert 8.0 < w["instructions_per_settled_cell"] < 10.0      # the committed SQ counter pass: 8.88
    # another configuration: the job figures are still the run's own, the committed instruction counts are not quoted
    w2 = bench.work_inflation(argparse.Namespace(grid=1024, queries=64, pipeline=4, tiled=False), 1, 1.0e6, c, passes=10)
    assert "jobs_per_touched_tile" in w2 and "instructions_per_settled_cell" not in w2
    # an older library without the counters (developer switch RNA_LIB): nothing invented
    assert bench.work_inflation(argparse.Namespace(grid=1024, queries=64, pipeline=4, tiled=False), 1, 1.0e6, None, passes=10) is None
