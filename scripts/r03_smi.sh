#!/bin/bash
# clocks and power while the bench runs (is the chip at its power limit?)
mkdir -p gpurun_out/r03
(RNA_BENCH_ONLY_ASTAR=${ONLY:-0} timeout 300 python bench.py --no-cpu --steps 150 > gpurun_out/r03/smi_bench.json 2>/dev/null) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|power\|busy" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/smi_bench.json").readline()); print(round(d["value"]), d["config"]["ms_per_pass"])
PY
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' '; echo "(idle)"
