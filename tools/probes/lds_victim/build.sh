#!/bin/bash
# cross-compiles the probe for gfx950 next to its source (the .so travels to the GPU box with the snapshot)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC lds_victim.hip -o liblds_victim.so
