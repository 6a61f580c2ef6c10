#!/bin/bash
# usage: tools/build_one_variant.sh <source stem, e.g. conv3x3_bf16> <out.so> [extra hipcc flags]
# rebuilds that one source with the flags and links it with the other objects of the in-tree build (python -m lct_vqa_amd.build first)
set -e
STEM=$1; OUT=$2; shift; shift
D=$(dirname $0)/../lct_vqa_amd/csrc
C=$(dirname $0)/../lct_vqa_amd/_C
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -Wno-inline-asm -I$D "$@" -c $D/$STEM.hip -o $T/$STEM.o
OBJS=$(ls $C/*.o | grep -v "/$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $T/$STEM.o $OBJS
rm -rf "$T"
