#!/bin/bash
# 1-rank RCCL rehearsal of the tensor-parallel launch sequence next to the single-GPU bench, same box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
timeout -k 10 400 python bench.py --steps 128 --warmup 16 --no-extras --no-cpu-baseline --no-traffic --no-per-kind 2>/dev/null | tail -1 > gpurun_out/r2_bench_single.json || exit 1
QEFT_BENCH_FORCE_TP=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 400 python bench.py --steps 128 --warmup 16 --no-extras --no-cpu-baseline --no-traffic --no-per-kind 2>/dev/null | tail -1 > gpurun_out/r2_bench_tp1.json || exit 1
python -c "
import json
a=json.load(open('gpurun_out/r2_bench_single.json')); b=json.load(open('gpurun_out/r2_bench_tp1.json'))
print('single', a['value'], 'tp1', b['value'], 'overhead %.1f %%' % (100 * (1 - b['value'] / a['value'])))"
done
