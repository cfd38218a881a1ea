#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/prof_prefill
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_prefill -o run --output-format csv -- python3 $R/tools/prefill_run.py > $R/gpurun_out/prof_prefill/log.txt 2>&1
echo rc=$?
cd $R
f=$(find gpurun_out/prof_prefill -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{r["Name"][:110]:110s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
grep prefill gpurun_out/prof_prefill/log.txt
