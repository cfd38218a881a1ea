#!/bin/bash
# GPU session 3: v3 (step-major, NW template) parity, then the variant lab, then the A/B bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gemv_v3.py tests/test_gpu_decode.py tests/test_gpu_engine_7b.py -q -s -p no:cacheprovider > gpurun_out/r2_t3.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r2_t3.log
grep -E "7b parity|passed|failed|FAILED|Error" gpurun_out/r2_t3.log | tail -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qeft_amd/csrc tools/gemv_v3_lab.hip -o gpurun_out/gemv_v3_lab && timeout -k 10 300 gpurun_out/gemv_v3_lab > gpurun_out/r2_v3_lab.txt 2>&1
echo "lab rc=$?"; cat gpurun_out/r2_v3_lab.txt
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --no-traffic > gpurun_out/r2_bench_v3b.json 2> gpurun_out/r2_bench_v3b.err
brc=$?
echo "bench v3 rc=$brc"; cut -c1-1600 gpurun_out/r2_bench_v3b.json; tail -n 5 gpurun_out/r2_bench_v3b.err
exit $brc
