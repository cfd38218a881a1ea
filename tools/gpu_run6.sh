#!/bin/bash
# GPU session: GEMM v3 parity, then ablations (loaders off / compute off) beside the normal run.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rc=0

echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/r2_t6.log | tail -30
if [ $rc -ne 0 ]; then exit $rc; fi
for a in 3 4 5; do
  echo "== QEFT_GEMM_ABL=$a" 
  QEFT_GEMM_ABL=$a timeout -k 10 200 python tools/bench_gemm.py --ms 2048 --reps 10 2>&1 | grep TFLOP || exit 1
done | tee gpurun_out/r2_gemm_v3_ablation.txt
