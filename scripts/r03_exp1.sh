#!/bin/bash
# where is the step bound now?  search capacity alone, reserved CUs, pipeline depth, bucket width
mkdir -p gpurun_out/r03
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 40 > gpurun_out/r03/exp1_$tag.json 2> gpurun_out/r03/exp1_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/exp1_$tag.json").readline())
    k=d["kernel_ms_per_pass"]
    print("$tag", round(d["value"]), "ms/pass %.3f"%d["config"]["ms_per_pass"], "search %.1f"%k["astar_search"], "engine %.2f"%sum(v for n,v in k.items() if n!="astar_search"), flush=True)
except Exception as ex: print("$tag failed", ex)
PY
}
run base X=1
run only RNA_BENCH_ONLY_ASTAR=1
run only_d10 RNA_BENCH_ONLY_ASTAR=1 RNA_ASTAR_PIPELINE=10
run only_skip0 RNA_BENCH_ONLY_ASTAR=1 RNA_SEARCH_CU_SKIP=8
run skip48 RNA_SEARCH_CU_SKIP=48
run skip64 RNA_SEARCH_CU_SKIP=64
run skip96 RNA_SEARCH_CU_SKIP=96
