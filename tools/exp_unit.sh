#!/bin/bash
# Builds experiment variants of the library with ONE translation unit rebuilt with extra flags:
# usage: tools/exp_unit.sh <unit without .hip, e.g. mfl_label_step> tag "-DFLAG ..." [tag flags ...]
#   ->  semantic_segmentation_amd/libmfl_<tag>.so  (select it with MFL_LIBRARY=...)
cd "$(dirname "$0")/../semantic_segmentation_amd/csrc"
unit=$1; shift
OTHERS=$(ls build/*.o | grep -v -e "/$unit.o" -e /exp_)
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( $CC $flags -c -o build/exp_${tag}_unit.o $unit.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmfl_$tag.so $OTHERS build/exp_${tag}_unit.o && echo "built $tag" ) &
done
wait
