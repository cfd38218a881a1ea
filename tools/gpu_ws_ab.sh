#!/bin/bash
# GPU box: parity of the weight-stationary tier, then its timing against ab/libqeft_hip_old.so (the previous build)
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -m gpu -k "17_to_64 or ragged_and_small or variant_names or 8_to_16" > gpurun_out/ws_tests.log 2>&1 || { tail -30 gpurun_out/ws_tests.log; exit 1; }
tail -3 gpurun_out/ws_tests.log
for i in 1 2; do
python tools/mid_m_time.py 24,32,48,64 > gpurun_out/ws_mid_m_new$i.txt 2>&1
QEFT_HIP_LIB=$PWD/ab/libqeft_hip_old.so python tools/mid_m_time.py 24,32,48,64 > gpurun_out/ws_mid_m_old$i.txt 2>&1
echo "--- new"; cat gpurun_out/ws_mid_m_new$i.txt; echo "--- old"; cat gpurun_out/ws_mid_m_old$i.txt
done
