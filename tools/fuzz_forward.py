#!/usr/bin/env python3
"""Random-shape parity sweep of the denoiser forward (HIP path) against the CPU oracle: python tools/fuzz_forward.py [N] [seed]
Draws N configurations (topology, J, d, heads, layers, B, T, guidance, compute dtype) from the ranges the kernels special-case
(tile shapes by row count, attention kernels by head width / sequence length / item count, the V2 window front end, K tails)
and prints every case with its error; exit code 1 if any exceeds its tolerance (fp32 2e-5, 16-bit modes 2e-2 of max|ref|).
Measurement / debugging aid: the CPU oracle is imported here as the checker, as in tests/."""
import os, random, sys, time
_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _root)
sys.path.insert(0, os.path.join(_root, "tests"))
import torch
from tests.test_gpu_parity import build_model, rel_err
from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs
from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
from oracle import mdm_forward as omf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
bad = 0
t00 = time.time()
for case in range(n):
    arch = rnd.choice(["mdm", "mdm_old"])
    dm = rnd.choice([128, 256, 512] + ([1024] if case % 7 == 0 else []))
    H = rnd.choice([h for h in (2, 4, 8) if dm // h in (32, 64, 128, 256)])      # gdx_create: head widths 32 / 64 / 128 / 256
    T = rnd.choice([10, 20, 30, 40, 60, 120, 200, 250]) if arch == "mdm" else rnd.choice([1, 7, 15, 16, 33, 64, 100, 196, 255, 300])
    J = rnd.choice([3, 16, 37, 150, 263, 498])
    L = rnd.choice([1, 2, 3])
    B = rnd.choice([1, 2, 3, 5, 9]) if T * dm <= 64 * 512 else rnd.choice([1, 2])
    dtype = rnd.choice(["fp32", "fp32", "fp16", "bf16"])
    cfgs = rnd.random() < 0.3
    cfg = dict(arch=arch, njoints=J, nfeats=1, latent_dim=dm, ff_size=rnd.choice([64, 192, 1024]), num_layers=L, num_heads=H, seed_poses=10)
    sd = init_state_dict(cfg, seed=case, perturb=True)
    m = build_model(arch, cfg, sd)
    m.compute_dtype = dtype
    x, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=100 + case)
    t = (torch.arange(B) * 97 + case) % 1000
    y = {"seed": seedp.to(dev), "mfcc": mfcc.to(dev)}
    with torch.no_grad():
        if cfgs:
            scale = torch.linspace(0.0, 2.5, B)
            y["scale"] = scale.to(dev)
            out = ClassifierFreeSampleModel(m)(x.to(dev), t.to(dev), y)
            c = omf.forward(sd, cfg, x, t, {"seed": seedp, "mfcc": mfcc})
            u = omf.forward(sd, cfg, x, t, {"seed": seedp, "mfcc": mfcc, "uncond": True})
            want = u + scale.view(-1, 1, 1, 1) * (c - u)
        else:
            out = m(x.to(dev), t.to(dev), y)
            want = omf.forward(sd, cfg, x, t, {"seed": seedp, "mfcc": mfcc})
    err = rel_err(out.cpu(), want)
    tol = 2e-5 if dtype == "fp32" else 2e-2
    ok = bool(torch.isfinite(out).all()) and err < tol
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} case {case:3d}: {arch:7s} J={J:3d} d={dm:4d} H={H} L={L} ff={cfg['ff_size']:4d} B={B} T={T:3d} "
          f"{'cfg ' if cfgs else '    '}{dtype}: rel err {err:.2e}  ({time.time() - t00:.0f} s)", flush=True)
print(f"{n - bad} / {n} within tolerance")
sys.exit(1 if bad else 0)
