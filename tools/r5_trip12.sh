#!/bin/bash
# Round-5 trip 12: kernel tables of the D = 16 sliced network and of a few chi = 64 MERA slices.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_rr16 $OUT/prof_mera64
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_rr16 -o rr -- python $OUT/../tools/rr64_probe.py --D 16 --max-slices 256 > $OUT/r5_rr16_prof.log 2>&1; echo "rr16 prof rc=$?")
python tools/kernel_stats.py $OUT/prof_rr16 "D = 16 64-node network, all 256 slices, staged reuse (tools/rr64_probe.py --D 16 --max-slices 256, warm-up + timed run)" > $OUT/r5_rr16_kernel_stats.txt; head -24 $OUT/r5_rr16_kernel_stats.txt
cat > /tmp/mera_slices.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["REPO"])
import tensornetwork_amd as ta
from tensornetwork_amd import workloads
be = ta.get_hip_backend()
layer = workloads.MeraSlicedLayer(be, 64, "left", ta.bfloat16)
sl = layer.all_slices()[:6]
st = {}
layer.contract(sl, stats=st); be.synchronize()
print(st)
PY
(cd /tmp && REPO=$OUT/.. timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mera64 -o mera -- python /tmp/mera_slices.py > $OUT/r5_mera64_prof.log 2>&1; echo "mera prof rc=$?")
python tools/kernel_stats.py $OUT/prof_mera64 "chi = 64 MERA placement 'left', the first 6 slices of the loop nest (i = 0, j = 0..5), staged reuse" > $OUT/r5_mera64_kernel_stats.txt; head -24 $OUT/r5_mera64_kernel_stats.txt
find $OUT/prof_rr16 $OUT/prof_mera64 -name "*kernel_trace.csv" -delete; rm -rf $OUT/prof_rr16 $OUT/prof_mera64
tail -2 $OUT/r5_rr16_prof.log; tail -2 $OUT/r5_mera64_prof.log
