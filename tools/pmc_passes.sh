cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SETS=${PMC_SETS:-"FETCH_SIZE|WRITE_SIZE TCC_HIT_sum TCC_MISS_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum|SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES|SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"}
IFS='|'
for set in $SETS; do
  unset IFS
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --only ${PMC_CFG:-c4} > $R/gpurun_out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/gpurun_out/pmc_$tag.log; exit 1; }
  echo "pass $tag ok"
done
