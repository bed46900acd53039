#!/bin/bash
# developer run (round 5): the fused map-update kernel in the default bench under several settings (workgroups, CU masks of the
# engine / side streams, CUs kept back from the searches)
# usage: bash scripts/r05_fused2.sh out_name
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/$1.txt
: > $OUT
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --steps 20 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']
    print('$label', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3), {n: round(v,3) for n,v in k.items() if n.startswith('himm') or n.startswith('compose') or n in ('vfh_step','astar_init')})
except Exception as ex:
    print('$label FAILED', ex)
" >> $OUT
  tail -n 2 /tmp/err.txt | grep -v amdgpu.ids >> $OUT
}
for rep in 1 2; do
  run "chain                      " RNA_HIMM_FUSED_WGS=0
  run "fused 128                  " RNA_HIMM_FUSED_WGS=128
  run "fused 112                  " RNA_HIMM_FUSED_WGS=112
  run "fused 96                   " RNA_HIMM_FUSED_WGS=96
  run "fused 128 eng+side mask 32 " RNA_HIMM_FUSED_WGS=128 RNA_ENGINE_CU_MASK=32 RNA_SIDE_CU_MASK=32
  run "fused 112 eng+side mask 32 " RNA_HIMM_FUSED_WGS=112 RNA_ENGINE_CU_MASK=32 RNA_SIDE_CU_MASK=32
  run "fused 96 skip24 masks 24   " RNA_HIMM_FUSED_WGS=96 RNA_SEARCH_CU_SKIP=24 RNA_ENGINE_CU_MASK=24 RNA_SIDE_CU_MASK=24
  run "fused 80 skip24 masks 24   " RNA_HIMM_FUSED_WGS=80 RNA_SEARCH_CU_SKIP=24 RNA_ENGINE_CU_MASK=24 RNA_SIDE_CU_MASK=24
done
cat $OUT
