#!/usr/bin/env python3
"""fp16 GEMM (csrc/gemmh.hip): correctness vs torch + timing.
python tools/gemmh_one.py M N K gelu [iters]      (env GDX_GEMMH_TILE=mb,nbw forces a tile)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gesturediffusion_amd import _lib
lib = _lib.load(); torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, gelu = (int(a) for a in sys.argv[1:5]); iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
if os.environ.get("CHECK", "1") == "1":
    Mc = min(M, int(os.environ.get("CHECK_M", "1000")))
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(Mc, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    C32 = torch.full((Mc, N), float("nan"), device="cuda"); C16 = torch.full((Mc, N), float("nan"), device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.gdx_linear_f16(vp(A), vp(W), vp(b), vp(C32), vp(C16), Mc, N, K, gelu, s), lib)
    ref = A.half().double() @ W.half().double().t() + b.double()
    if gelu: ref = torch.nn.functional.gelu(ref)
    e32 = ((C32.double() - ref).abs().max() / ref.abs().max()).item()
    e16 = ((C16.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"check M={Mc} N={N} K={K}: rel err fp32-out {e32:.2e}  fp16-out {e16:.2e}  nan={torch.isnan(C32).sum().item()}", flush=True)
us = C.c_float(); _lib.check(lib.gdx_bench_gemm_f16(M, N, K, gelu, iters, C.byref(us), s), lib)
tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12
print(f"M={M} N={N} K={K} gelu={gelu}: {us.value:.1f} us  {tf:.1f} TF  {tf/2500*100:.1f}% of 2.5 PF  env={ {k:v for k,v in os.environ.items() if k.startswith('GDX_')} }", flush=True)
