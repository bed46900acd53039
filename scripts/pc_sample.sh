#!/bin/bash
# developer run: PC sampling (rocprofv3, host-trap method) of a short default bench -- where the search kernel's wavefronts
# spend their time, by instruction.  Raw samples stay on the box; the histogram goes to gpurun_out/pcs/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pcs
rm -rf $OUT /tmp/pcs; mkdir -p $OUT /tmp/pcs
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${PCS_INTERVAL:-2000} \
  --kernel-trace --output-format csv -d /tmp/pcs -o pcs -- python3 $ROOT/bench.py --steps 6 --warmup 1 --no-cpu --no-check-paths > $OUT/run.log 2>&1
echo "rc $?" >> $OUT/run.log
find /tmp/pcs -type f | head -20 > $OUT/files.txt
for f in $(find /tmp/pcs -name "*pc_sampling*csv"); do
  echo "== $f $(wc -l < $f) lines" >> $OUT/files.txt
  head -4 $f >> $OUT/files.txt
  python3 $ROOT/scripts/pc_hist.py $f > $OUT/hist_$(basename $f).txt 2>&1
done
tail -5 $OUT/run.log; cat $OUT/files.txt | cut -c1-400 | head -40
