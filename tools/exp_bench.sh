#!/bin/bash
# Run bench.py once per experimental library build (semantic_segmentation_amd/libmfl_<TAG>.so).
for so in semantic_segmentation_amd/libmfl_?.so; do
  tag=$(basename $so .so)
  MFL_LIBRARY=$PWD/$so timeout -k 10 120 python -u bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/exp_$tag.log 2>&1
  echo "$tag exit=$? $(grep -o '"value": [0-9.]*' gpurun_out/exp_$tag.log | head -1) $(grep -o '"launch_ms": [0-9.]*' gpurun_out/exp_$tag.log)"
done
