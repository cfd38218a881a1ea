#!/bin/bash
# rocprofv3 kernel stats of the decode bench alone (no extras), optionally with an environment switch: ./tools/gpu_prof_decode.sh [VAR=VALUE] tag
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ $# -ge 2 ]; then export "$1"; TAG=$2; else TAG=${1:-default}; fi
OUT=$PWD/gpurun_out/prof_decode_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o run --output-format csv -- python3 $OLDPWD/bench.py --no-extras --no-cpu-baseline --no-traffic --no-per-kind --steps 128 --warmup 32 > $OUT/bench.json 2> $OUT/bench.err
cd $OLDPWD
python3 tools/summarize_rocprof.py $OUT/kt gpurun_out/prof_decode_$TAG.txt > /dev/null
rm -rf $OUT/kt
grep -E "gemv_v3_kernel|rope_attn" gpurun_out/prof_decode_$TAG.txt | head -8 | cut -c1-60,100-170
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'])"
