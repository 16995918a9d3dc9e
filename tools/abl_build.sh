#!/bin/bash
# Builds timing-experiment variants of one diagonal kernel (ABL_CH x ABL_KS, default 11 x 7):
# libmfl_<tag>.so = the regular objects with that kernel's object replaced.  usage: tools/abl_build.sh tag "-DFLAG ..." [tag flags ...]
cd "$(dirname "$0")/../semantic_segmentation_amd/csrc"
CH=${ABL_CH:-11}; KS=${ABL_KS:-7}
OTHERS=$(ls build/mfl_gather_*.o build/mfl_diag_*.o | grep -v mfl_diag_ch${CH}_k${KS}.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMFL_STAMPS -c -o build/abl_api_stamps.o mfl_api.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $ABL_API_FLAGS -c -o build/abl_api.o mfl_api.hip
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false -Wno-unused-function $flags \
      -DMFL_DIAG_CH=$CH -DMFL_DIAG_KS=$KS -c -o build/abl_$tag.o mfl_diag.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmfl_$tag.so $OTHERS $(case "$flags" in *MFL_STAMPS*) echo build/abl_api_stamps.o;; *) echo build/abl_api.o;; esac) build/abl_$tag.o && echo "built $tag" ) &
done
wait
