#!/bin/bash
# GPU box: build and run the persistent-chain lab (tools/chain_lab.hip) in several geometries; output -> gpurun_out/chain_lab_*.txt
set -e
mkdir -p gpurun_out build
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/chain_lab_build.log 2>&1
for cfg in ${CHAIN_CFGS:-"8 11" "16 5"}; do
  set -- $cfg
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLAB_NW=$1 -DLAB_D=$2 -I qeft_amd/csrc -I include -I tools tools/chain_lab.hip -L qeft_amd/lib -lqeft_hip -Wl,-rpath,$PWD/qeft_amd/lib -o build/chain_lab_$1_$2 >> gpurun_out/chain_lab_build.log 2>&1
  timeout -k 10 120 ./build/chain_lab_$1_$2 > gpurun_out/chain_lab_$1_$2.txt 2>&1 || true
  echo "=== NW=$1 D=$2"; grep -v "^\[epoch" gpurun_out/chain_lab_$1_$2.txt; grep -c "OK$" gpurun_out/chain_lab_$1_$2.txt || true
done
