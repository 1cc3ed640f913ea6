#!/bin/bash
# tools/build_igemm_variant.sh <name> <extra hipcc flags...>: libvnqa_<name>.so = the product library with csrc/conv_igemm.hip
# recompiled with extra flags (the other objects are reused from videonavqa_amd/lib/*.o) — timing-only A/B builds.
set -e
N=$1; shift
D=videonavqa_amd
mkdir -p /tmp/build/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -I include "$@" -c $D/csrc/conv_igemm.hip -o /tmp/build/$N/conv_igemm.hip.o
OBJS=$(ls $D/lib/*.o | grep -v conv_igemm.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib/libvnqa_$N.so /tmp/build/$N/conv_igemm.hip.o $OBJS
echo built $D/lib/libvnqa_$N.so
