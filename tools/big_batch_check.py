#!/usr/bin/env python3
"""Large-batch forward: rows of a very large batch against the same samples run as a batch of 2 (bit for bit in fp32, 2e-2 in
the 16-bit modes): python tools/big_batch_check.py  -- exercises the >2 GiB workspace tensors (32-bit buffer offsets)."""
import os, sys
_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests"))
import torch
from test_gpu_parity import build_model, rel_err, _real_cfg
from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs
dev = torch.device("cuda:0")
for arch, J, dm, T, B, dtype in (("mdm_old", 263, 512, 196, 2048, "fp32"), ("mdm_old", 263, 512, 196, 3000, "fp32"),
                                 ("mdm", 498, 1024, 520, 640, "fp16"), ("mdm_old", 263, 512, 196, 4096, "fp16")):
    cfg = _real_cfg(arch, J, dm)
    m = build_model(arch, cfg, init_state_dict(cfg, seed=0))
    m.compute_dtype = dtype
    x, seedp, mfcc = synthetic_inputs(cfg, 4, T, seed=10)
    idx = torch.arange(B) % 4
    xb, sb, mb = x[idx].to(dev), seedp[idx].to(dev), mfcc[idx].to(dev)
    t = torch.full((B,), 321, device=dev)
    try:
        full = m(xb, t, {"seed": sb, "mfcc": mb})
        sub = m(xb[:4], t[:4], {"seed": sb[:4], "mfcc": mb[:4]})
        torch.cuda.synchronize()
        errs = [rel_err(full[i:i + 4].cpu(), sub.cpu()) for i in (0, B // 2 // 4 * 4, B - 4)]
        print(f"{arch} d={dm} T={T} B={B} {dtype}: rows vs batch-of-4 rel err {errs}, finite={bool(torch.isfinite(full).all())}", flush=True)
    except Exception as e:
        print(f"{arch} d={dm} T={T} B={B} {dtype}: raised {type(e).__name__}: {str(e)[:200]}", flush=True)
    del m
    torch.cuda.empty_cache()
