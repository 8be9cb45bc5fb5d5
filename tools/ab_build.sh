#!/bin/bash
# Build an experimental variant of libsphx.so next to the product library:  tools/ab_build.sh NAME -DFOO=1 ...
# Run it with  SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/../yasph2d_amd/csrc"
name=$1; shift
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -Wno-unused-result "$@" -c sphx_kernels.hip -o /tmp/sphx_kernels_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libsphx_$name.so /tmp/sphx_kernels_$name.o sphx_host.o sphx_tiles.o -ldl -lpthread
echo built ../variants/libsphx_$name.so
