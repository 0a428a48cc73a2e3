#!/bin/bash
# builds tools/bw_probe/libbw_probe.so for gfx950 (cross-compiles without a GPU)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o libbw_probe.so bw_probe.hip
