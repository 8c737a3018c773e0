#!/usr/bin/env python3
"""One GEMM shape through gdx_bench_gemm: python tools/gemm_one.py M N K epi [iters]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gesturediffusion_amd import _lib
lib = _lib.load(); torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, epi = (int(a) for a in sys.argv[1:5]); iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
us = C.c_float(); _lib.check(lib.gdx_bench_gemm(M, N, K, epi, iters, C.byref(us), s), lib)
tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12
print(f"M={M} N={N} K={K} epi={epi}: {us.value:.1f} us  {tf:.1f} TF  {tf/157.3*100:.1f}%  env={ {k:v for k,v in os.environ.items() if k.startswith('GDX_')} }", flush=True)
