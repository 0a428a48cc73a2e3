#!/bin/bash
# Round 6, last session: tools/r6_final.sh (rocprofv3 stats + PMC passes -> traffic json, GPU suite, the driver's bench
# command, SVD kernel tables, MPS chain launches, SVD A/B probe) on the final tree, then the per-launch shape tables of
# the network workloads, the host-path probes and the k-major A/B of this session.
# usage: gpurun --timeout 5400 -- 'bash tools/r6b_final.sh'
set -u
bash tools/r6_final.sh
O=$PWD/gpurun_out/r6final
python tools/launch_shapes.py --workload mera --chi 32 --top 16 > $O/mera32_launch_shapes_after.txt 2>&1; head -5 $O/mera32_launch_shapes_after.txt
python tools/launch_shapes.py --workload mera64 --D 16 --top 12 > $O/mera64_launch_shapes_after.txt 2>&1; head -4 $O/mera64_launch_shapes_after.txt
python tools/launch_shapes.py --workload rr --D 12 --top 12 > $O/rr12_launch_shapes_after.txt 2>&1; head -3 $O/rr12_launch_shapes_after.txt
python tools/host_overhead_probe.py > $O/host_overhead_probe_after.txt 2>&1; head -2 $O/host_overhead_probe_after.txt; grep "mps chain" $O/host_overhead_probe_after.txt
python tools/host_path_ab.py --rows 32:L0,32:L1,64:L0 --rounds 2 > $O/host_path_ab.jsonl 2>&1; tail -2 $O/host_path_ab.jsonl
python tools/kmajor_lean_probe.py --rows 64,96,128,192,512row --iters 4 --out $O/kmajor_lean_probe.jsonl > /dev/null 2>&1; cut -c1-200 $O/kmajor_lean_probe.jsonl
