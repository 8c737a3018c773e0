#!/usr/bin/env python3
"""fp16 attention core (csrc/attentionh.hip): correctness vs torch (fp64 on the fp16-rounded inputs) + timing.
python tools/attnh_one.py B S H d"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gesturediffusion_amd import _lib
lib = _lib.load(); torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, S, H, d = (int(a) for a in sys.argv[1:5])
hd = d // H
Bc = min(B, 3)
g = torch.Generator(device="cuda").manual_seed(3)
qkv = torch.randn(Bc * S, 3 * d, device="cuda", generator=g)
qkv[:, :d] *= 2.0    # sharper softmax than unit-variance scores
ctx = torch.full((Bc * S, d), float("nan"), device="cuda")
_lib.check(lib.gdx_attention_f16(C.c_void_p(qkv.data_ptr()), C.c_void_p(ctx.data_ptr()), Bc, S, H, d, s), lib)
r = qkv.half().double().view(Bc, S, 3, H, hd)
q, k, v = (r[:, :, i].permute(0, 2, 1, 3) for i in range(3))
p = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1)
ref = (p @ v).permute(0, 2, 1, 3).reshape(Bc * S, d)
err = ((ctx.double() - ref).abs().max() / ref.abs().max()).item()
print(f"check B={Bc} S={S} H={H} d={d}: rel err {err:.2e}  nan={torch.isnan(ctx).sum().item()}", flush=True)
us = C.c_float(); _lib.check(lib.gdx_bench_attention(B, S, H, d, 3, 20, C.byref(us), s), lib)
fl = 4.0 * B * S * S * d
print(f"attention f16 B={B} S={S} H={H} d={d}: {us.value:.1f} us  {fl/(us.value*1e-6)/1e12:.1f} TF ({fl/(us.value*1e-6)/1e12/2500*100:.1f}% of 2.5 PF)", flush=True)
