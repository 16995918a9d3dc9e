#!/bin/bash
# Runs tools/abl_time.py once per libmfl_<tag>.so variant (on the GPU box).
cd "$(dirname "$0")/.."
for lib in semantic_segmentation_amd/libmfl_*.so; do
  MFL_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/abl_time.py 2>&1 | grep -v amdgpu.ids || exit 1
done
