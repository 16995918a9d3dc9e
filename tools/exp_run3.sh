#!/bin/bash
# On the GPU box: per libmfl_<tag>.so variant the step timing and two counter passes (L2 hits / misses, L1 read
# latency and pending stalls).  ABL_SHAPE selects the timed shape.
cd "$(dirname "$0")/.."
R=$PWD
for lib in semantic_segmentation_amd/libmfl_*.so; do
  tag=$(basename $lib .so)
  export MFL_LIBRARY=$R/$lib
  timeout -k 10 120 python3 tools/abl_time.py 2>&1 | grep -v amdgpu.ids || exit 1
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
    i=$((i + 1))
    ( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/py_$tag/pmc_$i -- python3 $R/tools/abl_time.py > $R/gpurun_out/py_${tag}_$i.log 2>&1 ) || { echo "pmc failed"; exit 1; }
  done
  python3 tools/pmc_summary.py gpurun_out/py_$tag ring_kernel
done
