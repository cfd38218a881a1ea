#!/bin/bash
# GPU session 2: v3 GEMV parity + engine parity, then A/B bench (round-1 launch sequence vs v3) on the same box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_gemv_v3.py tests/test_gpu_decode.py tests/test_gpu_engine_7b.py -q -s -p no:cacheprovider > gpurun_out/r2_t2.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r2_t2.log
grep -E "7b parity|passed|failed|FAILED|Error" gpurun_out/r2_t2.log | tail -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
QEFT_ENGINE_V2=1 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --no-traffic > gpurun_out/r2_bench_v2.json 2> gpurun_out/r2_bench_v2.err
echo "bench v2 rc=$?"; cut -c1-1500 gpurun_out/r2_bench_v2.json
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --no-traffic > gpurun_out/r2_bench_v3.json 2> gpurun_out/r2_bench_v3.err
brc=$?
echo "bench v3 rc=$brc"; cut -c1-1500 gpurun_out/r2_bench_v3.json; tail -n 5 gpurun_out/r2_bench_v3.err
exit $brc
