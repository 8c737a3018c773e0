#!/bin/bash
# fp16 attention: 8 waves x 2 blocks per launch-resident workgroup (q) vs the persistent form (p), same box, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for shape in "128 521 4 1024" "16 521 4 1024" "64 197 4 512" "256 197 4 512" "82 121 4 256"; do
  for v in q p; do
    GDX_ATTNH_WAVES=$v timeout -k 10 120 python tools/attnh_one.py $shape 2>&1 | grep -E "attention f16|err" | sed "s/$/ [$v]/" || exit 1
  done
done
done
