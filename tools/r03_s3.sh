#!/bin/bash
O=gpurun_out/r03_s3; mkdir -p $O; rm -f $O/attn.log
run() { echo "== $*" | tee -a $O/attn.log; timeout -k 10 120 "$@" >> $O/attn.log 2>&1; echo "rc=$?" >> $O/attn.log; }
for shape in "128 521 4 1024" "16 521 4 1024" "64 197 4 512" "256 197 4 512"; do
  run python tools/attnh_one.py $shape
  GDX_ATTNH8R=1 GDX_AH8R_ROT=0 run python tools/attnh_one.py $shape
  GDX_ATTNH8R=1 GDX_AH8R_ROT=1 run python tools/attnh_one.py $shape
done
grep -E "^==|check|attention f16|rc=[1-9]" $O/attn.log
