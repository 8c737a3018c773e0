#!/bin/bash
O=gpurun_out/r03_s4; mkdir -p $O
GDX_GEMM_DEBUG=1 GDX_ATTNH_WAVES=p timeout -k 10 120 python tools/attnh_one.py 128 521 4 1024 > $O/stamps_8p.log 2>&1
GDX_GEMM_DEBUG=1 GDX_ATTNH_WAVES=p timeout -k 10 120 python tools/attnh_one.py 16 521 4 1024 >> $O/stamps_8p.log 2>&1
grep -E "stamps|attention f16" $O/stamps_8p.log
