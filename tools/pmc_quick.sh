#!/bin/bash
# On the GPU box: memory-side counters of the fused step kernel for the library named by MFL_LIBRARY
# (two passes over tools/abl_time.py).  usage: bash tools/pmc_quick.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pq_${TAG}/pmc_$t -- python3 $R/tools/abl_time.py > $R/gpurun_out/pq_${TAG}_$t.log 2>&1 || { echo "pass $t failed"; tail -3 $R/gpurun_out/pq_${TAG}_$t.log; exit 1; }
done
echo "== $TAG"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pq_${TAG} ring_kernel
