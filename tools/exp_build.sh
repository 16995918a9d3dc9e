#!/bin/bash
# Builds experiment variants of the library: libmfl_<tag>.so = the regular objects with the C-ABI unit
# and one ring kernel unit (EXP_CH x EXP_KS, default 7 x 11) rebuilt with extra flags.
# usage: tools/exp_build.sh tag "-DFLAG ..." [tag flags ...]
cd "$(dirname "$0")/../semantic_segmentation_amd/csrc"
CH=${EXP_CH:-7}; KS=${EXP_KS:-11}
OTHERS=$(ls build/mfl_gather_*.o build/mfl_diag_*.o build/mfl_ring_*.o build/mfl_loop_backward.o build/mfl_label_step.o build/mfl_label_context.o build/mfl_preprocess.o | grep -v mfl_ring_${CH}_${KS}.o)
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( $CC $flags -c -o build/exp_${tag}_api.o mfl_api.hip && \
    $CC -fno-slp-vectorize $flags -DMFL_RING_CH=$CH -DMFL_RING_KS=$KS -c -o build/exp_${tag}_ring.o mfl_ring.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmfl_$tag.so $OTHERS build/exp_${tag}_api.o build/exp_${tag}_ring.o && echo "built $tag" ) &
done
wait
