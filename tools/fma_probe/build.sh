#!/bin/bash
# hipcc cross-compiles without a GPU; run the binary on the MI355X box (gpurun -- ./tools/fma_probe/fma_probe)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 fma_probe.hip -o fma_probe
