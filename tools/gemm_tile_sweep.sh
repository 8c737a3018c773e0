#!/bin/bash
# every tile shape of the persistent fp32 GEMM on one problem: tools/gemm_tile_sweep.sh M N K epi   (one process per shape: the override is read once)
M=$1; N=$2; K=$3; E=$4
echo "== M=$M N=$N K=$K epi=$E"
GDX_GEMM_DEBUG=1 python tools/gemm_one.py $M $N $K $E 50 2>&1 | grep -E "us |tile" | head -2
for t in 4,2,32 5,2,32 6,2,32 8,2,32 9,2,32 5,3,32 4,3,32 4,1,64 5,1,64 4,1,32 5,1,32 8,1,32 9,1,32 2,1,64 5,2,64 5,3,64 4,2,64 6,2,64 8,2,64 4,3,64; do
  GDX_GEMM_TILE=$t python tools/gemm_one.py $M $N $K $E 50 2>&1 | grep -E "us " | cut -c1-90
done
