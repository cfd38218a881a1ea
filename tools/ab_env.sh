#!/bin/bash
# A/B of an environment switch on ONE box, same library: ./tools/ab_env.sh VAR=VALUE  (decode bench without extras, alternating)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for tag in off on; do
    if [ $tag = on ]; then export "$1"; else unset "${1%%=*}"; fi
    timeout -k 10 300 python bench.py --steps 128 --warmup 32 --no-extras --no-cpu-baseline --no-traffic --no-per-kind 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $tag', d['value'], d['ms_per_step'], d['config']['last_token'])" || exit 1
  done
done
