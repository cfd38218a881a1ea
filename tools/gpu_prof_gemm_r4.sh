#!/bin/bash
# round 4: the PMC passes of tools/gpu_prof_gemm.sh over the three GEMM tiers (256 x 128 tiles, 128 x 128 tiles, gemm_ws)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
: > gpurun_out/r04_gemm_pmc.txt
for args in "4096 4096 2048" "4096 4096 1024" "11008 4096 2048" "4096 4096 64"; do
  echo "==== N K M = $args" >> gpurun_out/r04_gemm_pmc.txt
  GEMM_ARGS="$args" bash tools/gpu_prof_gemm.sh >> gpurun_out/r04_gemm_pmc.txt 2>&1
  rm -rf gpurun_out/prof_gemm
done
grep -v "^\"void at::\|distribution_elementwise" gpurun_out/r04_gemm_pmc.txt | cut -c1-160
