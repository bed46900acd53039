#!/bin/bash
mkdir -p gpurun_out/r03
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 > gpurun_out/r03/e3_$tag.json 2> gpurun_out/r03/e3_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/e3_$tag.json").readline())
    k=d["kernel_ms_per_pass"]
    print("$tag", round(d["value"]), "ms/pass %.3f"%d["config"]["ms_per_pass"], "search %.1f"%k["astar_search"], "engine %.2f"%sum(v for n,v in k.items() if n not in ("astar_search","astar_reset")), {n:round(v,2) for n,v in k.items()}, flush=True)
except Exception as ex: print("$tag failed", ex, open("gpurun_out/r03/e3_$tag.err").read()[-300:])
PY
}
timeout 300 python -m pytest tests -m gpu -x -q -k "astar" 2>&1 | tail -2
run base X=1
run pad_skip0 RNA_TSA_LDS_PAD=12600 RNA_SEARCH_CU_SKIP=0
run pad_skip8 RNA_TSA_LDS_PAD=12600 RNA_SEARCH_CU_SKIP=8
run pad_skip16 RNA_TSA_LDS_PAD=12600 RNA_SEARCH_CU_SKIP=16
run skip16 RNA_SEARCH_CU_SKIP=16
run skip48 RNA_SEARCH_CU_SKIP=48
