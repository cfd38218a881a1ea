#!/bin/bash
# GPU box: shared-GPU rehearsals of bench.py with more ranks / the 13B shapes (one-shot collective; every rank on cuda:0)
mkdir -p gpurun_out
for cfg in "4 7b" "2 13b" "4 13b"; do
  set -- $cfg
  timeout -k 10 500 python bench.py --gpus $1 --model $2 --steps 16 --warmup 4 --no-extras > gpurun_out/rehearse_tp$1_$2.json.log 2> gpurun_out/rehearse_tp$1_$2.err || { echo "FAILED tp$1 $2"; tail -30 gpurun_out/rehearse_tp$1_$2.err; exit 1; }
  python - gpurun_out/rehearse_tp$1_$2.json.log <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
m = r["multi_gpu"]
print(sys.argv[1], r["value"], r["n_gpus"], r["config"]["hipgraph"], m["collective"], m.get("collective_note"), m["collective_us"])
PY
done
