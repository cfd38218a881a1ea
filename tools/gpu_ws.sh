#!/bin/bash
# GPU box: parity of the weight-stationary tier + its timing
set -e
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -m gpu -k "17_to_64 or ragged_and_small or variant_names or 8_to_16" > gpurun_out/ws_tests.log 2>&1 || { tail -30 gpurun_out/ws_tests.log; exit 1; }
tail -3 gpurun_out/ws_tests.log
python tools/mid_m_time.py 16,24,32,48,64,128 > gpurun_out/ws_mid_m.txt 2>&1
QEFT_GEMM_WS=0 python tools/mid_m_time.py 24,32,48,64 > gpurun_out/ws_mid_m_off.txt 2>&1
cat gpurun_out/ws_mid_m.txt; echo "--- QEFT_GEMM_WS=0"; cat gpurun_out/ws_mid_m_off.txt
