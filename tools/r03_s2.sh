#!/bin/bash
# round 3, session 2: first runs of the one-wave-per-SIMD fp16 attention (attentionh4.hip) against the 8-wave kernels
O=gpurun_out/r03_s2; mkdir -p $O
run() { echo "== $*" | tee -a $O/attn.log; timeout -k 10 120 "$@" >> $O/attn.log 2>&1; echo "rc=$?" >> $O/attn.log; }
run python tools/attnh_one.py 128 521 4 1024
GDX_ATTNH4=1 GDX_AH4_QBMAX=1 run python tools/attnh_one.py 128 521 4 1024
GDX_ATTNH4=1 GDX_AH4_QBMAX=2 run python tools/attnh_one.py 128 521 4 1024
GDX_ATTNH4=1 run python tools/attnh_one.py 128 521 4 1024
run python tools/attnh_one.py 16 521 4 1024
GDX_ATTNH4=1 GDX_AH4_QBMAX=2 run python tools/attnh_one.py 16 521 4 1024
GDX_ATTNH4=1 run python tools/attnh_one.py 16 521 4 1024
run python tools/attnh_one.py 64 197 4 512
GDX_ATTNH4=1 GDX_AH4_QBMAX=3 run python tools/attnh_one.py 64 197 4 512
GDX_ATTNH4=1 run python tools/attnh_one.py 64 197 4 512
run python tools/attnh_one.py 256 197 4 512
GDX_ATTNH4=1 run python tools/attnh_one.py 256 197 4 512
grep -E "^==|check|attention f16" $O/attn.log
