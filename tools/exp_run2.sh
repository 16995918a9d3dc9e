#!/bin/bash
# On the GPU box: per libmfl_<tag>.so variant an 11x11 parity check (local forward and two fused steps against
# the oracle), the step timing, and one counter pass (L1 pending stalls).  ABL_SHAPE selects the timed shape.
cd "$(dirname "$0")/.."
R=$PWD
for lib in semantic_segmentation_amd/libmfl_*.so; do
  tag=$(basename $lib .so)
  export MFL_LIBRARY=$R/$lib
  if [ -n "$EXP_CHECK" ]; then
    timeout -k 10 120 python3 tools/check_ring.py 1,19,96,160,11 2>&1 | grep -v amdgpu.ids || exit 1
    timeout -k 10 120 python3 tools/check_mf.py 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
  fi
  timeout -k 10 120 python3 tools/abl_time.py 2>&1 | grep -v amdgpu.ids || exit 1
  if [ -n "$EXP_PMC" ]; then
    ( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/px_$tag/pmc_1 -- python3 $R/tools/abl_time.py > $R/gpurun_out/px_$tag.log 2>&1 ) || { echo "pmc failed"; exit 1; }
    python3 tools/pmc_summary.py gpurun_out/px_$tag ring_kernel
  fi
done
