#!/usr/bin/env python3
"""Latency of a whole sampling loop at small batch (serving shape): python tools/small_loop.py [dtype] [B] [T] [J] [d] [steps]
Set GDX_NO_GRAPH=1 to disable the hipGraph replay of the step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import build_model
from gesturediffusion_amd.diffusion import gaussian_diffusion as gd
from gesturediffusion_amd.diffusion.respace import SpacedDiffusion, space_timesteps
from gesturediffusion_amd.utils.init import synthetic_inputs
dt = sys.argv[1] if len(sys.argv) > 1 else "fp16"
B, T, J, d, steps = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((2, 4), (3, 60), (4, 150), (5, 512), (6, 1000)))
dev = torch.device("cuda:0")
model, cfg, sd = build_model("mdm", J, d, 8, dev)
model.compute_dtype = dt
df = SpacedDiffusion(use_timesteps=space_timesteps(1000, [steps]), betas=gd.get_named_beta_schedule("cosine", 1000),
                     model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE)
_, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=10)
y = {"seed": seedp.to(dev), "mfcc": mfcc.to(dev)}
run = lambda: df.p_sample_loop(model, (B, J, 1, T), clip_denoised=False, model_kwargs={"y": y}, rng="philox", philox_seed=1)
a = run(); torch.cuda.synchronize()
t0 = time.perf_counter(); b = run(); torch.cuda.synchronize(); el = time.perf_counter() - t0
assert torch.equal(a, b) and torch.isfinite(a).all()
print(f"{dt} B={B} T={T} J={J} d={d}: {steps}-step loop {el*1e3:.1f} ms = {el*1e3/steps:.4f} ms/step  graph={'off' if os.environ.get('GDX_NO_GRAPH') else 'on'}", flush=True)
