#!/bin/bash
# builds tools/energy_probe/libenergy_probe.so for gfx950 (cross-compiles without a GPU)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o libenergy_probe.so energy_probe.hip
