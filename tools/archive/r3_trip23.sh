#!/bin/bash
# band SVD: per-panel kernel durations along the panel sequence
set -u
O=gpurun_out/${1:-r3t23}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o band4096 -- python $GRAFT_REPO_ROOT/tools/svd_band_probe.py 4096 256 gauss --no-check > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $O/prof/band4096_kernel_trace.csv | grep -v "mean=   0.0\|mean=  -0.0" | tee $O/trace_summary.txt
rm -f $O/prof/*kernel_trace.csv
