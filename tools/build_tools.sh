#!/bin/bash
# standalone HIP microbenchmarks of tools/ -> tools/bin/ (git-ignored; travels to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
for t in ${@:-dma_bench store_bench}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/$t tools/$t.hip
  echo built tools/bin/$t
done
