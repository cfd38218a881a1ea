#!/bin/bash
# GPU box: TP tests + a 2-rank shared-GPU rehearsal of bench.py with the one-shot collective
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_tp_processes.py tests/test_gpu_tp.py -x -q -m gpu > gpurun_out/tp_tests.log 2>&1 || { tail -30 gpurun_out/tp_tests.log; exit 1; }
tail -2 gpurun_out/tp_tests.log
timeout -k 10 600 python bench.py --gpus 2 --model 7b --steps 16 --warmup 4 --no-extras > gpurun_out/bench_tp2_oneshot.json.log 2> gpurun_out/bench_tp2_oneshot.err || { tail -20 gpurun_out/bench_tp2_oneshot.err; exit 1; }
tail -c 3000 gpurun_out/bench_tp2_oneshot.json.log
timeout -k 10 600 python bench.py --gpus 2 --model 7b --steps 16 --warmup 4 --no-extras --collective rccl > gpurun_out/bench_tp2_gloo.json.log 2> gpurun_out/bench_tp2_gloo.err || { tail -20 gpurun_out/bench_tp2_gloo.err; exit 1; }
tail -c 1500 gpurun_out/bench_tp2_gloo.json.log
