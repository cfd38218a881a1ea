#!/bin/bash
# GPU box: the tensor-parallel process tests (gloo, ipc one-shot) + the TP unit tests
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_tp_processes.py tests/test_gpu_tp.py -x -q -m gpu > gpurun_out/tp_tests.log 2>&1
rc=$?
tail -25 gpurun_out/tp_tests.log
exit $rc
