#!/bin/bash
# GPU box: GEMM routing change -- the tests that assert routing tiers, then the M sweep new vs ab/libqeft_hip_old.so
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_gemm.py tests/test_gpu_w3.py tests/test_gpu_zz_alloc_guard.py tests/test_gpu_engine_7b.py -q -m gpu > gpurun_out/route_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/route_tests.log | tail -15
MS=512,640,768,896,1024,1152,1280,1408,1536,1664,1792,1920,2048
python tools/mid_m_time.py $MS 2>/dev/null > gpurun_out/r04_gemm_m_sweep_new.txt
QEFT_HIP_LIB=$PWD/ab/libqeft_hip_old.so python tools/mid_m_time.py $MS 2>/dev/null > gpurun_out/r04_gemm_m_sweep_old.txt
echo "--- new"; cat gpurun_out/r04_gemm_m_sweep_new.txt; echo "--- old"; cat gpurun_out/r04_gemm_m_sweep_old.txt
