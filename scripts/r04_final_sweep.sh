#!/bin/bash
# developer run: reserved CUs, pipeline depth and bucket width once more on the kernel with the row scan
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/r04_sweep_env.sh r04_sweep_reserved_cus_scan.txt RNA_SEARCH_CU_SKIP 24 28 32 36 40
for p in 12 13 14 15; do R04_ARGS="--pipeline $p" bash scripts/r04_sweep_env.sh r04_tmp.txt RNA_X $p > /dev/null; sed "s/^RNA_X=/--pipeline /" gpurun_out/r04_tmp.txt | cut -c1-100; done > gpurun_out/r04_sweep_pipeline_depth_scan.txt
for b in 48000 96000 144000; do R04_ARGS="--bucket-width $b" bash scripts/r04_sweep_env.sh r04_tmp.txt RNA_X $b > /dev/null; sed "s/^RNA_X=/--bucket-width /" gpurun_out/r04_tmp.txt | cut -c1-100; done > gpurun_out/r04_sweep_bucket_width_scan.txt
rm -f gpurun_out/r04_tmp.txt
cat gpurun_out/r04_sweep_pipeline_depth_scan.txt gpurun_out/r04_sweep_bucket_width_scan.txt
