#!/bin/bash
mkdir -p gpurun_out/r03
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 > gpurun_out/r03/e4_$tag.json 2> gpurun_out/r03/e4_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/e4_$tag.json").readline())
    k=d["kernel_ms_per_pass"]
    print("$tag", round(d["value"]), "ms/pass %.3f"%d["config"]["ms_per_pass"], "search %.1f"%k["astar_search"], "engine %.2f"%sum(v for n,v in k.items() if n not in ("astar_search","astar_reset")), {n:round(v,2) for n,v in k.items()}, flush=True)
except Exception as ex: print("$tag failed", ex, open("gpurun_out/r03/e4_$tag.err").read()[-300:])
PY
}
run m32 RNA_ENGINE_CU_MASK=32
run m48 RNA_ENGINE_CU_MASK=48 RNA_SEARCH_CU_SKIP=48
run m64 RNA_ENGINE_CU_MASK=64 RNA_SEARCH_CU_SKIP=64
run m40 RNA_ENGINE_CU_MASK=40 RNA_SEARCH_CU_SKIP=40
run m32_d15 RNA_ENGINE_CU_MASK=32 RNA_ASTAR_PIPELINE=14
