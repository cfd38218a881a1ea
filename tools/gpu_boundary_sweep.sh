#!/bin/bash
# GPU box: the boundary_gemv record under the lab overrides of the ring depth (QEFT_GEMV_DEPTH)
mkdir -p gpurun_out
: > gpurun_out/boundary_sweep.txt
for d in 0 2 4; do
  echo "== QEFT_GEMV_DEPTH=$d" >> gpurun_out/boundary_sweep.txt
  if [ $d = 0 ]; then timeout -k 10 200 python tools/boundary_time.py 2>/dev/null >> gpurun_out/boundary_sweep.txt
  else QEFT_GEMV_DEPTH=$d timeout -k 10 200 python tools/boundary_time.py 2>/dev/null >> gpurun_out/boundary_sweep.txt; fi
done
python - <<'PY'
import re, ast
for line in open("gpurun_out/boundary_sweep.txt"):
    line = line.strip()
    if line.startswith("=="):
        print(line); continue
    shape, d = line.split(" ", 1)
    d = ast.literal_eval(d)
    print("  ", shape, {k.split("_")[0]: v[0] for k, v in d.items() if k.endswith("gemv_4bit_qeft")})
PY
