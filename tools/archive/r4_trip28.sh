#!/bin/bash
# D = 12 network on the final tree: kernel table and the kernels of one slice in launch order.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t28; mkdir -p $O
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof_final -o rr -- python $GRAFT_REPO_ROOT/tools/rr64_probe.py --D 12 --max-slices 4 > $O/prof_final.log 2>&1; echo "prof rc=$?")
python tools/kernel_stats.py $O/prof_final "rocprofv3 --kernel-trace --stats -- python tools/rr64_probe.py --D 12 --max-slices 4 (warm-up + timed pass: 8 slices; final tree: gather lowering + K loop + wider tail split; MI355X, round 4)" > $O/rr64_D12_final_kernel_stats.txt 2>&1
python tools/kernel_seq.py $O/prof_final --slices 8 --min-us 40 > $O/rr64_D12_final_kernel_seq.txt 2>&1; cat $O/rr64_D12_final_kernel_seq.txt
