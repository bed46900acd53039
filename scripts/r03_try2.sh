#!/bin/bash
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests -m gpu -x -q -k "astar" > gpurun_out/r03/pytest_astar.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03/pytest_astar.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 40 > gpurun_out/r03/t2_$tag.json 2> gpurun_out/r03/t2_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/t2_$tag.json").readline())
    k=d["kernel_ms_per_pass"]
    print("$tag", round(d["value"]), "ms/pass %.3f"%d["config"]["ms_per_pass"], "search %.1f"%k["astar_search"], "engine %.2f"%sum(v for n,v in k.items() if n!="astar_search"), flush=True)
except Exception as ex: print("$tag failed", ex)
PY
}
run base X=1
run only RNA_BENCH_ONLY_ASTAR=1
RNA_LIB=librna_stats.so RNA_BENCH_ONLY_ASTAR=1 timeout 300 python bench.py --no-cpu --steps 20 2>&1 | grep "tsa stats"
