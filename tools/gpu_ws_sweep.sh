#!/bin/bash
# GPU box: the weight-stationary tier's geometry sweep (QEFT_WS_MC x QEFT_WS_RSC), M = 32 and 64 on the three 7B shapes
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
: > gpurun_out/ws_sweep.txt
for mc in 1 2; do for rsc in 1 2 3 4 5 6; do
  echo "== MC=$mc RSC=$rsc" >> gpurun_out/ws_sweep.txt
  QEFT_WS_MC=$mc QEFT_WS_RSC=$rsc python tools/mid_m_time.py 24,32,48,64 2>/dev/null >> gpurun_out/ws_sweep.txt
done; done
cat gpurun_out/ws_sweep.txt
