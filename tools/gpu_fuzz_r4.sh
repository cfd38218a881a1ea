#!/bin/bash
# GPU box: the seeded parity sweeps on the final round-4 build (as round 3's, plus the gemm_ws mode)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rc=0
timeout -k 10 900 python tools/fuzz_parity.py 150 > gpurun_out/r04_fuzz_parity_small.txt 2>&1 || rc=1; tail -1 gpurun_out/r04_fuzz_parity_small.txt
timeout -k 10 900 python tools/fuzz_parity.py 120 large > gpurun_out/r04_fuzz_parity_large.txt 2>&1 || rc=1; tail -1 gpurun_out/r04_fuzz_parity_large.txt
timeout -k 10 900 python tools/fuzz_parity.py 80 v3 > gpurun_out/r04_fuzz_parity_v3.txt 2>&1 || rc=1; tail -1 gpurun_out/r04_fuzz_parity_v3.txt
timeout -k 10 900 python tools/fuzz_parity.py 160 ref > gpurun_out/r04_fuzz_parity_ref.txt 2>&1 || rc=1; tail -1 gpurun_out/r04_fuzz_parity_ref.txt
timeout -k 10 900 python tools/fuzz_parity.py 40 w3gemm > gpurun_out/r04_fuzz_parity_w3gemm.txt 2>&1 || rc=1; tail -1 gpurun_out/r04_fuzz_parity_w3gemm.txt
timeout -k 10 900 python tools/fuzz_parity.py 120 ws > gpurun_out/r04_fuzz_parity_ws.txt 2>&1 || rc=1; tail -1 gpurun_out/r04_fuzz_parity_ws.txt
exit $rc
