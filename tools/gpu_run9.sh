#!/bin/bash
# GPU session: tensor-parallel engine tests, then the 1-rank RCCL rehearsal of the multi-GPU launch sequence next to the single-GPU bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tp.py -q -p no:cacheprovider -x > gpurun_out/r2_t9.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/r2_t9.log | tail -30
if [ $rc -ne 0 ]; then tail -60 gpurun_out/r2_t9.log; exit $rc; fi
timeout -k 10 400 python bench.py --steps 128 --warmup 16 --no-extras --no-cpu-baseline --no-traffic > gpurun_out/r2_bench_single.json 2> gpurun_out/r2_bench_single.err
echo "single rc=$?"; tail -c 1500 gpurun_out/r2_bench_single.json
QEFT_BENCH_FORCE_TP=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 400 python bench.py --steps 128 --warmup 16 --no-extras --no-cpu-baseline --no-traffic > gpurun_out/r2_bench_tp1.json 2> gpurun_out/r2_bench_tp1.err
echo "tp1 rc=$?"; tail -c 1500 gpurun_out/r2_bench_tp1.json; tail -5 gpurun_out/r2_bench_tp1.err
