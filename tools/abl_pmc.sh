#!/bin/bash
# Instruction counts per ablation variant (on the GPU box): rocprofv3 --pmc over tools/abl_time.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in $R/semantic_segmentation_amd/libmfl_*.so; do
  tag=$(basename $lib .so)
  MFL_LIBRARY=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/ablpmc_$tag -- python3 $R/tools/abl_time.py > $R/gpurun_out/ablpmc_$tag.log 2>&1 || { echo "$tag failed"; exit 1; }
  echo "$tag ok"
done
