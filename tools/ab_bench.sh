#!/bin/bash
# A/B of two builds of the library on ONE box: QEFT_HIP_LIB selects the .so (qeft_amd/lib/libqeft_hip_{old,new}.so);
# alternating runs of the decode bench without extras.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for tag in old new; do
    QEFT_HIP_LIB=$PWD/qeft_amd/lib/libqeft_hip_$tag.so timeout -k 10 300 python bench.py --steps 128 --warmup 32 --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_launch'], {k:v['us'] for k,v in d['roofline']['per_launch_kind'].items()})" || exit 1
  done
done
