#!/bin/bash
# rocprofv3 kernel trace of a 2048-token prefill of the whole 7B model (tools/prefill_run.py); prints the kernels of the LAST
# of its four passes (model construction and the cold first pass are in the trace too) -> gpurun_out/r2_prefill_kernels.txt
cd "$(dirname "$0")/.."
R=$PWD
rm -rf gpurun_out/prof_prefill; mkdir -p gpurun_out/prof_prefill
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_prefill -o run --output-format csv -- python3 $R/tools/prefill_run.py > $R/gpurun_out/prof_prefill/log.txt 2>&1
echo rc=$?
cd $R
python3 - > gpurun_out/r2_prefill_kernels.txt <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/prof_prefill/run_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = [i for i, r in enumerate(rows) if 'gemm_w4_kernel_v3' in r['Kernel_Name']]
sel = rows[g[-(len(g) // 4)] - 3:]            # the last of the four passes: its GEMM launches and what lies between them
t0 = int(sel[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in sel)
acc = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r['Kernel_Name'][:96]
    acc[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); acc[k][1] += 1
busy = sum(v[0] for v in acc.values())
print(f"last prefill pass (2048 tokens, Llama-2-7B w4 g128 r128): {(t1 - t0) / 1e6:.2f} ms first kernel start -> last kernel end, {len(sel)} kernels, {busy / 1e6:.2f} ms busy")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:96s} {v[1]:5d} {v[0] / 1e6:8.3f} ms {v[0] / v[1] / 1e3:8.1f} us {100 * v[0] / busy:5.1f} %")
PY
cat gpurun_out/r2_prefill_kernels.txt; grep prefill gpurun_out/prof_prefill/log.txt
