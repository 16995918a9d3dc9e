#!/bin/bash
# Builds experiment variants of the library with the label-context unit rebuilt with extra flags:
# usage: tools/exp_context.sh tag "-DFLAG ..." [tag flags ...]  ->  semantic_segmentation_amd/libmfl_<tag>.so
cd "$(dirname "$0")/../semantic_segmentation_amd/csrc"
OTHERS=$(ls build/*.o | grep -v -e mfl_label_context.o -e /exp_)
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( $CC $flags -c -o build/exp_${tag}_ctx.o mfl_label_context.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmfl_$tag.so $OTHERS build/exp_${tag}_ctx.o && echo "built $tag" ) &
done
wait
