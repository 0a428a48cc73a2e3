#!/bin/bash
# rocprofv3 kernel stats of the sliced random-regular network (per-slice kernel mix).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
R=$PWD
cd /tmp
for D in ${RR_DS:-12 8}; do
  rm -rf $OUT/prof_stats_rr$D
  rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_rr$D -o rr -- python $R/tools/rr64_probe.py --D $D --min-slices 64 --max-slices ${RR_SLICES:-16} > $OUT/rr$D.log 2>&1
  tail -1 $OUT/rr$D.log
done
cd $R
python tools/prof_summary.py gpurun_out gpurun_out/rr_summary.txt > /dev/null
grep -A14 "prof_stats_rr" gpurun_out/rr_summary.txt
