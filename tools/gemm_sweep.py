#!/usr/bin/env python3
"""GEMM micro-benchmark sweep through the C ABI (gdx_bench_gemm): TFLOP/s and fraction of the
fp32 MFMA peak for the denoiser's GEMM shapes and for large-K probes (mainloop-only efficiency)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gesturediffusion_amd import _lib

lib = _lib.load()
torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(12608, 1536, 512, 0, "qkv"), (12608, 512, 512, 0, "out-proj"), (12608, 1024, 512, 1, "ffn1+gelu"),
          (12608, 1024, 512, 0, "ffn1 bias only"), (12608, 512, 1024, 0, "ffn2"), (12800, 1024, 512, 0, "M=12800"),
          (16384, 1024, 512, 0, "M=16384 (128 tiles x 8)"), (16384, 2048, 4096, 0, "big K"), (8192, 8192, 8192, 0, "8k^3"),
          (25216, 1024, 512, 1, "cfg ffn1")]
for M, N, K, epi, name in shapes:
    us = C.c_float()
    _lib.check(lib.gdx_bench_gemm(M, N, K, epi, 20, C.byref(us), s), lib)
    tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:5d} epi={epi}  {us.value:9.1f} us  {tf:7.1f} TF  {tf / 157.3 * 100:5.1f}% of f32 MFMA peak", flush=True)
