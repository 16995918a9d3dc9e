#!/bin/bash
# Builds timing-experiment variants of the (CH=11, k=7) diagonal kernel: libmfl_<tag>.so = the regular
# objects with mfl_diag_ch11_k7.o replaced.  usage: tools/abl_build.sh tag "-DFLAG ..." [tag flags ...]
cd "$(dirname "$0")/../semantic_segmentation_amd/csrc"
OTHERS=$(ls build/mfl_gather_*.o build/mfl_diag_*.o | grep -v mfl_diag_ch11_k7.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMFL_STAMPS -c -o build/abl_api_stamps.o mfl_api.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c -o build/abl_api.o mfl_api.hip
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false -Wno-unused-function $flags \
      -DMFL_DIAG_CH=11 -DMFL_DIAG_KS=7 -c -o build/abl_$tag.o mfl_diag.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmfl_$tag.so $OTHERS $(case "$flags" in *MFL_STAMPS*) echo build/abl_api_stamps.o;; *) echo build/abl_api.o;; esac) build/abl_$tag.o && echo "built $tag" ) &
done
wait
