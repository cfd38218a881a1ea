#!/bin/bash
# GPU session: the -m gpu suite, then (only if pytest ended normally: rc 0 or 1) the default bench line.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_t1.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r2_t1.log
tail -n 15 gpurun_out/r2_t1.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
brc=$?
echo "bench rc=$brc"
tail -c 3000 gpurun_out/r2_bench1.json
tail -n 5 gpurun_out/r2_bench1.err
exit $brc
