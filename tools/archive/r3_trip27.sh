#!/bin/bash
set -u
O=gpurun_out/${1:-r3t27}
mkdir -p $O
rm -f $O/repro4.txt
echo "--- env KPHASE=0" | tee -a $O/repro4.txt
TNH_GEMM_KPHASE=0 timeout 300 python - < tools/_cubes_body.py 2>&1 | tail -8 | tee -a $O/repro4.txt
echo "--- env FOO_BAR_BAZ_QUX=0" | tee -a $O/repro4.txt
FOO_BAR_BAZ_QUX=0 timeout 300 python - < tools/_cubes_body.py 2>&1 | tail -8 | tee -a $O/repro4.txt
echo "--- env KPHASE=1" | tee -a $O/repro4.txt
TNH_GEMM_KPHASE=1 timeout 300 python - < tools/_cubes_body.py 2>&1 | tail -8 | tee -a $O/repro4.txt
echo "--- env KPHASE=0 serialize" | tee -a $O/repro4.txt
AMD_SERIALIZE_KERNEL=3 TNH_GEMM_KPHASE=0 timeout 300 python - < tools/_cubes_body.py 2>&1 | tail -8 | tee -a $O/repro4.txt
