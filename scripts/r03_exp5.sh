#!/bin/bash
mkdir -p gpurun_out/r03
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 > gpurun_out/r03/e5_$tag.json 2> gpurun_out/r03/e5_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/e5_$tag.json").readline())
    k=d["kernel_ms_per_pass"]
    print("$tag", round(d["value"]), "ms/pass %.3f"%d["config"]["ms_per_pass"], "search %.1f"%d["kernel_ms_per_pass_timed_region"]["astar_search"], "engine(profiled turn) %.2f"%sum(v for n,v in k.items() if n not in ("astar_search","astar_reset")), flush=True)
except Exception as ex: print("$tag failed", ex, open("gpurun_out/r03/e5_$tag.err").read()[-300:])
PY
}
run base X=1
run skip16 RNA_SEARCH_CU_SKIP=16
run skip24 RNA_SEARCH_CU_SKIP=24
run skip48 RNA_SEARCH_CU_SKIP=48
run d12 RNA_ASTAR_PIPELINE=12
run d14 RNA_ASTAR_PIPELINE=14
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 | cut -c1-120
