#!/bin/bash
# On the GPU box: wave-level counters of the kernels of tools/time_local.py (or $PMC_SCRIPT) for the library
# named by MFL_LIBRARY.  usage: bash tools/pmc_sq.sh <tag> <kernel name pattern>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; PAT=${2:-ring_kernel}
SCRIPT=${PMC_SCRIPT:-tools/time_local.py}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sq_${TAG}/pmc_$t -- python3 $R/$SCRIPT > $R/gpurun_out/sq_${TAG}_$t.log 2>&1 || { echo "pass $t failed"; tail -3 $R/gpurun_out/sq_${TAG}_$t.log; exit 1; }
done
echo "== $TAG $PAT"; python3 $R/tools/pmc_summary.py $R/gpurun_out/sq_${TAG} "$PAT"
