#!/bin/bash
# GPU session: the whole -m gpu suite, smoke(), then the default bench (with PMC child) -- what the driver runs at round end.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/r4_full_tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" gpurun_out/r4_full_tests.log | tail -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then tail -40 gpurun_out/r4_full_tests.log; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1
echo "smoke rc=$?"; tail -5 gpurun_out/r4_smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo "bench rc=$?"; tail -c 6000 gpurun_out/r4_bench_default.json; tail -5 gpurun_out/r4_bench_default.err
