#!/bin/bash
# On the GPU box: the profiles.  Kernel traces of the bench per configuration, counter passes of the
# headline configuration (19 labels, 1024x2048, 11x11) and of the 21-label 512x512 configuration, counters of
# the single operators (backward), and the FETCH_SIZE calibration.  Every rocprofv3 run is a run of its own.
# Usage (repository root on the box): [ROUND=r03] bash tools/profile_all.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r03}
O=$R/gpurun_out/$ROUND
mkdir -p $O
trace() {   # tag, bench arguments
  local tag=$1; shift
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/trace_$tag.log 2>&1 || { echo "trace $tag failed"; tail -5 $O/trace_$tag.log; exit 1; }
  echo "trace $tag ok"
}
trace c4 --only c4 --steps 3 --warmup 1
trace c2 --only c2 --steps 3 --warmup 1
trace c5 --only c5 --steps 2 --warmup 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ops -- python3 $R/tools/time_local.py > $O/trace_ops.log 2>&1 || { echo "trace ops failed"; exit 1; }
echo "trace ops ok"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k50 -- python3 $R/tools/time_k50.py > $O/trace_k50.log 2>&1 || { echo "trace k50 failed"; exit 1; }
echo "trace k50 ok"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ctx -- python3 $R/tools/time_context.py > $O/trace_ctx.log 2>&1 || { echo "trace ctx failed"; exit 1; }
echo "trace ctx ok"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -- python3 $R/tools/time_train.py > $O/trace_train.log 2>&1 || { echo "trace train failed"; exit 1; }
echo "trace train ok"
SETS="FETCH_SIZE|WRITE_SIZE TCC_HIT_sum TCC_MISS_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum|SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES|SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
pmc() {   # tag, program and arguments
  local tag=$1; shift
  local i=0
  IFS='|'
  for set in $SETS; do
    unset IFS
    i=$((i + 1))
    timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag/pmc_$i -- "$@" > $O/pmc_${tag}_$i.log 2>&1 || { echo "pmc $tag pass $i failed"; tail -3 $O/pmc_${tag}_$i.log; exit 1; }
    IFS='|'
  done
  unset IFS
  echo "pmc $tag ok"
}
pmc c4 python3 $R/bench.py --only c4 --steps 1 --warmup 0 --no-cpu-baseline
pmc c2 python3 $R/bench.py --only c2 --steps 1 --warmup 0 --no-cpu-baseline
pmc c5 python3 $R/bench.py --only c5 --steps 1 --warmup 0 --no-cpu-baseline
pmc ops python3 $R/tools/time_local.py
pmc ctx python3 $R/tools/time_context.py
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib -- $R/tools/exp/exp_calib_fetch > $O/pmc_calib.log 2>&1 || { echo "calibration failed"; tail -5 $O/pmc_calib.log; exit 1; }
echo "calibration ok"
