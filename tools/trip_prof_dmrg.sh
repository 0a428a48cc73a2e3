#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m cProfile -o gpurun_out/dmrg.prof tools/dmrg_probe.py --n 16 --bonds 128 --cpu-max 0 | tail -2
python -c "
import pstats
p = pstats.Stats('gpurun_out/dmrg.prof'); p.sort_stats('tottime').print_stats(40)
" 2>&1 | tail -60
