"""Turns rocprofv3's rocpd SQLite outputs (gpurun_out/prof_rNN/...) into the text/JSON summaries
committed under profiles/.  Usage: python scripts/rocpd_summary.py gpurun_out/prof_r01 profiles r01"""
import json
import os
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    return [dict(name=r[0], calls=r[1], total_ns=r[2], avg_ns=r[3], min_ns=r[4], max_ns=r[5], pct=100.0 * r[2] / total)
            for r in rows]


def pmc_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                     "group by kernel_name, counter_name order by sum(value) desc").fetchall()
    return [dict(name=r[0], counter=r[1], dispatches=r[2], sum=r[3], avg=r[4]) for r in rows]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    out = {}
    for sub, fn in (("trace", "bench_results.db"), ("trace_p1", "bench_p1_results.db"), ("trace_rows", "rows_results.db")):
        p = os.path.join(src, sub, fn)
        if os.path.exists(p):
            st = kernel_stats(p)
            out[sub] = st
            with open(os.path.join(dst, "%s_kernel_stats_%s.txt" % (tag, sub)), "w") as f:
                f.write("# rocprofv3 --kernel-trace --stats -- python3 %s ... (%s)\n" % ("scripts/bench_rows.py" if sub == "trace_rows" else "bench.py", sub))
                f.write("%-34s %6s %14s %14s %14s %14s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
                for r in st:
                    f.write("%-34s %6d %14d %14.0f %14d %14d %7.2f\n" % (short(r["name"]), r["calls"], r["total_ns"], r["avg_ns"],
                                                                         r["min_ns"], r["max_ns"], r["pct"]))
    pmc = {}
    for sub, fn in (("pmc_fetch", "fetch_results.db"), ("pmc_write", "write_results.db")):
        p = os.path.join(src, sub, fn)
        if os.path.exists(p):
            for r in pmc_stats(p):
                pmc.setdefault(short(r["name"]), {})[r["counter"]] = dict(dispatches=r["dispatches"], avg=r["avg"])
    if pmc:
        # FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B (rocprofv3); on gfx950 FETCH_SIZE counts half of
        # the bytes of wide streaming reads (MI355X_MICROARCH.md, HBM section) -> doubled for the byte estimate
        summ = {}
        for k, v in pmc.items():
            f = v.get("FETCH_SIZE", {}).get("avg")
            w = v.get("WRITE_SIZE", {}).get("avg")
            summ[k] = dict(fetch_size_kb_avg=f, write_size_kb_avg=w,
                           hbm_bytes_per_launch=(2.0 * (f or 0.0) + (w or 0.0)) * 1024.0)
        out["pmc"] = summ
        with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as f:
            json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py "
                                "--steps 2 --warmup 1 --no-cpu` (r02: --steps 8 --warmup 4; the default configuration below); per launch: "
                                "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE doubled: gfx950 "
                                "reports half of the bytes of 16 B/lane streaming reads, MI355X_MICROARCH.md; dword "
                                "accesses are uncalibrated, so bench.py quotes the range from (FETCH + WRITE) * 1024)",
                           config=dict(grid=4096, queries=256, pipeline=bench_pipeline(src), ray_poses=64, rays_per_pose=1563),
                           kernels=summ), f, indent=1)
    print(json.dumps({k: (v if k == "pmc" else [(short(r["name"]), r["calls"], round(r["avg_ns"] / 1e3, 1)) for r in v][:8])
                      for k, v in out.items()}, indent=1)[:3000])


def bench_pipeline(src):
    """pipeline depth of the profiled bench runs, from the JSON line the PMC pass printed"""
    for name in ("pmc_fetch.log", "bench_under_rocprof.log"):
        try:
            for line in open(os.path.join(src, name)):
                if line.startswith("{") and "astar_pipeline_depth" in line:
                    return json.loads(line)["config"]["astar_pipeline_depth"]
        except OSError:
            pass
    return None


if __name__ == "__main__":
    main()
