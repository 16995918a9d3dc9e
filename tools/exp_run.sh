#!/bin/bash
# On the GPU box: per libmfl_<tag>.so variant, a parity check against the oracle and the step timing.
cd "$(dirname "$0")/.."
for lib in semantic_segmentation_amd/libmfl_*.so; do
  echo "== $lib"
  MFL_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/check_diag.py 2>&1 | grep -v amdgpu.ids || exit 1
  MFL_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/abl_time.py 2>&1 | grep -v amdgpu.ids || exit 1
done
