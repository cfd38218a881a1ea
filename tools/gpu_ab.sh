#!/bin/bash
# A/B of two builds of the library on one GPU box: the in-tree build vs build/ab/libqeft_hip_old.so (QEFT_HIP_LIB);
# GEMV / decode parity tests of the in-tree build first.  Output -> gpurun_out/ab_*
set -o pipefail
mkdir -p gpurun_out
OLD=$PWD/ab/libqeft_hip_old.so
timeout -k 10 900 python -m pytest tests/test_gpu_gemv.py tests/test_gpu_gemv_v3.py tests/test_gpu_decode.py tests/test_gpu_engine_7b.py tests/test_gpu_w3.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for tag in new old new old; do
  if [ $tag = old ]; then export QEFT_HIP_LIB=$OLD; else unset QEFT_HIP_LIB; fi
  timeout -k 10 400 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/ab_bench_$tag.log 2>gpurun_out/ab_bench_$tag.err || { tail -20 gpurun_out/ab_bench_$tag.err; exit 1; }
  python - $tag <<'PY'
import json, sys
tag = sys.argv[1]
r = json.loads(open(f"gpurun_out/ab_bench_{tag}.log").read().strip().splitlines()[-1])
k = r["roofline"].get("per_launch_kind", {})
print(tag, r["value"], r["roofline"]["frac"], {n: v.get("us") for n, v in k.items()} if isinstance(k, dict) else k)
b = r.get("boundary_gemv", {}).get("per_shape", [])
for s in b:
    print("   ", s["shape"], {n.split("_")[0]: v["us"] for n, v in s.items() if n.endswith("gemv_4bit_qeft")})
PY
done
