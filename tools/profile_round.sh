#!/bin/bash
# On the GPU box: kernel trace of the bench, the PMC passes, and the FETCH_SIZE calibration.
# Usage (from the repository root on the box): bash tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1 || { echo "kernel trace failed"; tail -5 $R/gpurun_out/prof_$TAG.log; exit 1; }
echo "kernel trace ok"
bash $R/tools/pmc_passes.sh || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_calib -- $R/tools/exp/exp_calib_fetch > $R/gpurun_out/pmc_calib.log 2>&1 || { echo "calibration failed"; tail -5 $R/gpurun_out/pmc_calib.log; exit 1; }
echo "calibration ok"
