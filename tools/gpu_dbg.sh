#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python tools/dbg_hn.py > gpurun_out/dbg_new.txt 2>&1; tail -25 gpurun_out/dbg_new.txt
QEFT_HIP_LIB=$PWD/ab/libqeft_hip_old.so timeout -k 10 200 python tools/dbg_hn.py > gpurun_out/dbg_old.txt 2>&1; tail -25 gpurun_out/dbg_old.txt
