#!/bin/bash
# On the GPU box: vector-memory-pipeline counters (address unit, L1, translation) of the kernels of
# tools/abl_time.py (or $PMC_SCRIPT) for the library named by MFL_LIBRARY.  One rocprofv3 run per pass.
# usage: bash tools/pmc_mem.sh <tag> <kernel name pattern>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; PAT=${2:-ring_kernel}
SCRIPT=${PMC_SCRIPT:-tools/abl_time.py}
i=0
for set in \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE" \
  "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM" \
  "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TCP_LATENCY_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"; do
  i=$((i + 1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pm_${TAG}/pmc_$i -- python3 $R/$SCRIPT > $R/gpurun_out/pm_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pm_${TAG}_$i.log; exit 1; }
done
echo "== $TAG $PAT"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pm_${TAG} "$PAT"
