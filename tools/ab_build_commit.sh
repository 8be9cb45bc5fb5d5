#!/bin/bash
# Build libsphx from the kernel sources of a COMMIT as a variant library:  tools/ab_build_commit.sh NAME COMMIT [-DFOO ...]
# (same-box A/B against the working tree's library: SPHX_LIB=yasph2d_amd/variants/libsphx_NAME.so)
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; commit=$2; shift 2
d=/tmp/ab_src_$name; rm -rf $d; mkdir -p $d
for f in sphx_kernels.hip sphx_launch.inc sphx_internal.hpp sphx_host.hpp sphx_host.cpp sphx_tiles.cpp; do git -C $root show $commit:yasph2d_amd/csrc/$f > $d/$f; done
mkdir -p $d/../include_$name; cp -r $root/include $d/..  2>/dev/null || true
cd $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -Wno-unused-result -I$root/include -I$root/yasph2d_amd/csrc"
/opt/rocm/bin/hipcc $F "$@" -c sphx_kernels.hip -o k.o
mkdir -p $root/yasph2d_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/yasph2d_amd/variants/libsphx_$name.so k.o $root/yasph2d_amd/csrc/sphx_host.o $root/yasph2d_amd/csrc/sphx_tiles.o -ldl -lpthread
echo built yasph2d_amd/variants/libsphx_$name.so from $commit
