#!/bin/bash
# builds the round-2 hc kernels as they were BEFORE commit c161093 (LDS float atomics in hc_bwd's gradient flush) into a
# stand-alone library with the same C ABI; the sources r02_* are frozen copies of this repository's own files at c161093^
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I . -shared -x hip r02_hc.hip r02_plan.hip -o libhc_r02.so
