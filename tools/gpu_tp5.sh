#!/bin/bash
# GPU box: 2-rank shared-GPU rehearsal of bench.py, normal and with a faked give-up of the one-shot collective (fallback path)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tp_processes.py tests/test_gpu_tp.py tests/test_gpu_gemv.py -x -q -m gpu -k "tp or oneshot or two_rank or shadow" > gpurun_out/tp_tests.log 2>&1 || { tail -30 gpurun_out/tp_tests.log; exit 1; }
tail -2 gpurun_out/tp_tests.log
for fake in 0 1; do
  QEFT_BENCH_FAKE_ONESHOT_GIVEUP=$fake timeout -k 10 600 python bench.py --gpus 2 --model 7b --steps 16 --warmup 4 --no-extras > gpurun_out/tp5_$fake.log 2> gpurun_out/tp5_$fake.err || { echo FAILED fake=$fake; tail -30 gpurun_out/tp5_$fake.err; exit 1; }
  python - $fake <<'PY'
import json, sys
r = json.loads(open(f"gpurun_out/tp5_{sys.argv[1]}.log").read().strip().splitlines()[-1])
m = r["multi_gpu"]
print("fake", sys.argv[1], "->", r["value"], "graph", r["config"]["hipgraph"], m["collective"], m.get("collective_note"), m["collective_us"])
PY
done
