#!/bin/bash
# usage: tools/build_w4_variant.sh <out.so> [extra hipcc flags]  -- rebuilds only conv3x3_wino4.hip, links the other objects
set -e
OUT=$1; shift
D=$(dirname $0)/../lct_vqa_amd/csrc
C=$(dirname $0)/../lct_vqa_amd/_C
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -Wno-inline-asm -I$D "$@" -c $D/conv3x3_wino4.hip -o $T/conv3x3_wino4.o
OBJS=$(ls $C/*.o | grep -v conv3x3_wino4.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $T/conv3x3_wino4.o $OBJS
rm -rf "$T"
