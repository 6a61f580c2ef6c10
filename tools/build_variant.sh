#!/bin/bash
# usage: tools/build_variant.sh <out.so> [extra hipcc flags, e.g. -DCONV_WBUFS=2]
set -e
OUT=$1; shift
D=$(dirname $0)/../lct_vqa_amd/csrc
T=$(mktemp -d)
for f in api conv3x3 conv3x3_bwd conv3x3_wino darts gemm ops; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -Wno-inline-asm -I$D "$@" -c $D/$f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $T/*.o
rm -rf $T
