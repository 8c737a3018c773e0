#!/bin/bash
# fp16 GEMM tile sweep on the denoiser's shapes: tools/gemmh_sweep.sh > gpurun_out/ghsweep.log
cd "$(dirname "$0")/.."
export CHECK=0
for shape in "66688 3072 1024 0" "66688 1024 1024 0" "66688 1024 1024 1" "12608 1536 512 0" "12608 512 512 0" "12608 1024 512 1" "12608 512 1024 0"; do
  for t in 4,1 8,1 4,2 8,2 4,4 8,4; do
    GDX_GEMMH_TILE=$t timeout -k 10 120 python tools/gemmh_one.py $shape 20 2>&1 | grep "TF" || exit 1
  done
done
