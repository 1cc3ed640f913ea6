#!/bin/bash
# tools/build_one_variant.sh <name> <source.hip> <extra hipcc flags...>: libvnqa_<name>.so = the product library with ONE source
# recompiled with extra flags (the other objects are reused from videonavqa_amd/lib/*.o) — timing-only A/B builds.
set -e
N=$1; SRC=$2; shift 2
D=videonavqa_amd
mkdir -p /tmp/build/$N
PF=""
if [ "$SRC" = "conv_wreg.hip" ] || [ "$SRC" = "conv_ps.hip" ]; then PF="-mllvm -pragma-unroll-threshold=4000000 -Werror=pass-failed"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -I include $PF "$@" -c $D/csrc/$SRC -o /tmp/build/$N/$SRC.o
OBJS=$(ls $D/lib/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib/libvnqa_$N.so /tmp/build/$N/$SRC.o $OBJS
echo built $D/lib/libvnqa_$N.so
