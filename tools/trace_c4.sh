#!/bin/bash
# On the GPU box: kernel trace of tools/time_c4.py; prints the average duration of the stream kernels' launches.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tc4
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tc4 -- python3 $R/tools/time_c4.py > $R/gpurun_out/tc4.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/tc4.log; exit 1; }
grep -v amdgpu.ids $R/gpurun_out/tc4.log | tail -4
f=$(find $R/gpurun_out/tc4 -name "*kernel_stats.csv" | sort | tail -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "ring" in n or "weights" in n or "pad_" in n:
        print(f'{n[:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
