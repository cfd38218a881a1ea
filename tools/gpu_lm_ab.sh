#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
python tools/lm_head_time.py 2>/dev/null | head -1 | sed 's/^/new  /'
QEFT_HIP_LIB=$PWD/ab/libqeft_hip_old.so python tools/lm_head_time.py 2>/dev/null | head -1 | sed 's/^/old  /'
done
timeout -k 10 300 python -m pytest tests/test_gpu_decode.py -x -q -m gpu -k "token or lm_head or norm or engine" 2>&1 | tail -2
