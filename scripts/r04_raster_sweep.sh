#!/bin/bash
# developer run: the map-update chain alone on 32 CUs (RNA_ENGINE_CU_MASK=32) and on the whole chip, for several builds
# of the rasteriser and numbers of workgroups it is launched with
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for lib in "$@"; do
  for wgs in 0 128 256 512; do
    for mask in 32 0; do
      export RNA_ENGINE_CU_MASK=$mask RNA_LIB=$lib
      if [ $wgs -gt 0 ]; then export RNA_HIMM_RASTER_WGS=$wgs; else unset RNA_HIMM_RASTER_WGS; fi
      echo "$lib wgs=$wgs mask=$mask: $(python3 scripts/himm_alone.py 300 2>/dev/null | tail -1)"
    done
  done
done
