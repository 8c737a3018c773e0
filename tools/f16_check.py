#!/usr/bin/env python3
"""fp16 mode vs the reference-generated fp32 golden outputs at the BASELINE shapes (prints relative errors)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden, rel_err
from test_gpu_parity import build_model, _real_cfg
from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs
g = load_golden("real_shapes.npz")
d = torch.device("cuda:0")
for name, arch, J, dm in [("c1_v2", "mdm", 150, 512), ("c2_v1", "mdm_old", 263, 512), ("c2_v2", "mdm", 263, 512), ("c5_v2", "mdm", 498, 1024)]:
    B, T = int(g[name + ".meta"][0]), int(g[name + ".meta"][1])
    cfg = _real_cfg(arch, J, dm)
    sd = init_state_dict(cfg, seed=0)
    x, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=10)
    t = torch.from_numpy(g[name + ".t"]).to(d)
    y = {"seed": seedp.to(d), "mfcc": mfcc.to(d)}
    res = {}
    for dt in ("fp32", "fp16"):
        m = build_model(arch, cfg, sd); m.compute_dtype = dt
        out = m(x.to(d), t, y); out_u = m(x.to(d), t, dict(y, uncond=True))
        res[dt] = (rel_err(out.cpu(), g[name + ".out"]), rel_err(out_u.cpu(), g[name + ".out_uncond"]))
    print(f"{name}: B={B} T={T}  fp32 {res['fp32'][0]:.2e}/{res['fp32'][1]:.2e}   fp16 {res['fp16'][0]:.2e}/{res['fp16'][1]:.2e}", flush=True)
