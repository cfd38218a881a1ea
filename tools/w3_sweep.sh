#!/bin/bash
# 3-bit decode on one box: waves per block / ring depth overrides of the v3 GEMV against the defaults and against 4 bits
cd "$(dirname "$0")/.."
run() { timeout -k 10 300 python bench.py --bits $1 --steps 128 --warmup 32 --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bits $1 NW=${QEFT_GEMV_NW:-auto} D=${QEFT_GEMV_DEPTH:-auto}', d['value'], d['roofline']['us_per_launch'], {k: v['us'] for k, v in d['roofline']['per_launch_kind'].items()})"; }
run 4 || exit 1
run 3 || exit 1
QEFT_GEMV_NW=16 run 3 || exit 1
QEFT_GEMV_NW=8 run 3 || exit 1
QEFT_GEMV_DEPTH=4 run 3 || exit 1
QEFT_GEMV_NW=16 QEFT_GEMV_DEPTH=4 run 3 || exit 1
run 4
