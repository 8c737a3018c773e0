#!/usr/bin/env python3
"""Attention core micro-benchmark: python tools/attn_one.py B S H d [version]
(version 1 = attention.hip, 3 = fp16 attentionh.hip, 4 = attention3.hip; default: 1 and 4)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gesturediffusion_amd import _lib
lib = _lib.load(); torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, S, H, d = (int(a) for a in sys.argv[1:5])
for ver in ([int(sys.argv[5])] if len(sys.argv) > 5 else [1, 4]):
    us = C.c_float(); _lib.check(lib.gdx_bench_attention(B, S, H, d, ver, 20, C.byref(us), s), lib)
    fl = 4.0 * B * S * S * d
    print(f"attention v{ver} B={B} S={S} H={H} d={d}: {us.value:.1f} us  {fl/(us.value*1e-6)/1e12:.1f} TF ({fl/(us.value*1e-6)/1e12/157.3*100:.1f}% of f32 MFMA peak)", flush=True)
