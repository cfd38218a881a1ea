#!/bin/bash
# A/B of two builds on one GPU box, decode line only: in-tree build vs ab/libqeft_hip_old.so (3 alternating runs each)
mkdir -p gpurun_out
OLD=$PWD/ab/libqeft_hip_old.so
for i in 1 2 3; do for tag in new old; do
  if [ $tag = old ]; then export QEFT_HIP_LIB=$OLD; else unset QEFT_HIP_LIB; fi
  timeout -k 10 300 python bench.py --no-traffic --no-cpu-baseline --no-extras > gpurun_out/abq_$tag.log 2>gpurun_out/abq_$tag.err || { tail -20 gpurun_out/abq_$tag.err; exit 1; }
  python - $tag <<'PY'
import json, sys
r = json.loads(open(f"gpurun_out/abq_{sys.argv[1]}.log").read().strip().splitlines()[-1])
k = r["roofline"].get("per_launch_kind", {})
print(sys.argv[1], r["value"], r["roofline"]["frac"], {n: v.get("us") for n, v in k.items()})
PY
done; done
