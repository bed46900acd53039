#!/bin/bash
# developer run (round 5): the fused map-update kernel -- HIMM / compose / tiled parity tests with it, then the default bench with
# the chain of separate launches (RNA_HIMM_FUSED_WGS=0) and with the fused kernel at several workgroup counts
# usage: bash scripts/r05_fused.sh out_name [wgs ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05
NAME=$1; shift
if [ "${R05_TESTS:-1}" = "1" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiled.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not astar and not rrt and not vfh_config2" > gpurun_out/r05/${NAME}_tests.txt 2>&1
  tail -n 6 gpurun_out/r05/${NAME}_tests.txt
fi
OUT=gpurun_out/r05/${NAME}.txt
: > $OUT
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --steps 20 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']
    print('$label', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3), {n: round(v,3) for n,v in k.items() if n.startswith('himm') or n.startswith('compose')})
except Exception as ex:
    print('$label FAILED', ex)
" >> $OUT
  tail -n 2 /tmp/err.txt | grep -v amdgpu.ids >> $OUT
}
for rep in 1 2; do
  run "chain        " RNA_HIMM_FUSED_WGS=0
  for w in "$@"; do run "fused wgs=$w " RNA_HIMM_FUSED_WGS=$w; done
done
cat $OUT
