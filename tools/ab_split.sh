#!/bin/bash
# fp16 GEMM row split (main rounds of 256x256 tiles + small-tile tail) vs the unsplit kernel, same box, interleaved
cd "$(dirname "$0")/.."
export GDX_GEMM_DEBUG_OFF=1
CHECK_M=66688 timeout -k 10 120 python tools/gemmh_one.py 66688 1024 1024 1 5 2>&1 | grep -E "check|TF" || exit 1
export CHECK=0
for rep in 1 2; do
for shape in "66688 1024 1024 0" "66688 1024 1024 1" "66688 3072 1024 0"; do
  GDX_GEMMH_NOSPLIT=1 timeout -k 10 120 python tools/gemmh_one.py $shape 30 2>&1 | grep "TF" | sed 's/$/ [whole]/' || exit 1
  timeout -k 10 120 python tools/gemmh_one.py $shape 30 2>&1 | grep "TF" | sed 's/$/ [split]/' || exit 1
done
done
