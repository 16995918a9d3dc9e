#!/bin/bash
# Builds timing-experiment variants of one diagonal kernel (ABL_CH x ABL_KS, default 11 x 7):
# libmfl_<tag>.so = the regular objects with that kernel's object replaced.  usage: tools/abl_build.sh tag "-DFLAG ..." [tag flags ...]
cd "$(dirname "$0")/../semantic_segmentation_amd/csrc"
CH=${ABL_CH:-11}; KS=${ABL_KS:-7}
OTHERS=$(ls build/mfl_api.o build/mfl_loop_backward.o build/mfl_label_step.o build/mfl_label_context.o build/mfl_gather_*.o build/mfl_diag_ch*.o build/mfl_ring_[0-9]*.o | grep -v "mfl_diag_ch${CH}_k${KS}.o")
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false -Wno-unused-function $flags \
      -DMFL_DIAG_CH=$CH -DMFL_DIAG_KS=$KS -c -o build/abl_$tag.o mfl_diag.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmfl_$tag.so $OTHERS build/abl_$tag.o && echo "built $tag" ) &
done
wait
