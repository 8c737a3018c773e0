#!/usr/bin/env python3
"""fp32 attention3 kernel at the BASELINE shapes through gdx_bench_attention (version 4): python tools/attn3_bench.py
(round 2: requesting the second K/V tile together with the first measured 57.2 vs 56.4 us at config 2, 23.69 vs 23.83 at the
GENEA chunk shape: not kept)"""
import ctypes as C, torch, sys
sys.path.insert(0, ".")
from gesturediffusion_amd import _lib
lib = _lib.load(); torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B,S,H,d) in [(64,197,4,512),(82,121,4,256),(512,197,4,512),(4,61,4,512),(64,201,4,512)]:
    us = C.c_float()
    for rep in range(2):
        _lib.check(lib.gdx_bench_attention(B,S,H,d,4,300,C.byref(us),s), lib)
    print("attention3", B,S,H,d, round(us.value,2), "us")
