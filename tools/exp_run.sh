#!/bin/bash
# On the GPU box: the step timing per libmfl_<tag>.so variant (EXP_CHECK=1: parity check first).
cd "$(dirname "$0")/.."
for lib in semantic_segmentation_amd/libmfl_*.so; do
  if [ -n "$EXP_CHECK" ]; then MFL_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/check_ring.py 2>&1 | grep -v amdgpu.ids || exit 1; fi
  MFL_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/abl_time.py 2>&1 | grep -v amdgpu.ids || exit 1
done
