#!/usr/bin/env python3
"""Random-option parity sweep of the fused sampling loop (gdx_sample_loop, noise tape) against the CPU oracle's loop:
python tools/fuzz_loops.py [N] [seed].  Draws topology, J, T (token-major and general loop state), B, schedule length, ancestral /
DDIM (eta 0 / 0.5), guidance, clip_denoised, const_noise, skip_timesteps, init_image, inpainting, dump_steps and the compute
dtype; exit code 1 if any case exceeds its tolerance (fp32 2e-4, 16-bit modes 2e-2 of max|ref|).  The oracle is the checker."""
import os, random, sys, time
_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests"))
import torch
from test_gpu_parity import build_model, rel_err, _diffusion
from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs
from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
from oracle import mdm_forward as omf, sampler as osamp, schedule as osch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
bad, t00 = 0, time.time()
for case in range(n):
    arch = rnd.choice(["mdm", "mdm_old"])
    J = rnd.choice([16, 18, 37])
    T = rnd.choice([20, 30, 40]) if arch == "mdm" else rnd.choice([12, 18, 20, 33, 36])
    B = rnd.choice([1, 2, 3])
    steps = rnd.choice([6, 12])
    kind = rnd.choice(["p", "p", "ddim"])
    eta = rnd.choice([0.0, 0.5]) if kind == "ddim" else 0.0
    cfgs, clip = rnd.random() < 0.4, rnd.random() < 0.4
    const = kind == "p" and rnd.random() < 0.25
    skip = rnd.choice([0, 0, 2])
    use_init = rnd.random() < 0.3
    inpaint = rnd.random() < 0.25
    dump = [0, steps - skip - 1] if kind == "p" and rnd.random() < 0.2 else None
    dtype = rnd.choice(["fp32", "fp32", "fp16", "bf16"])
    cfg = dict(arch=arch, njoints=J, nfeats=1, latent_dim=128, ff_size=256, num_layers=2, num_heads=4, seed_poses=10)
    sd = init_state_dict(cfg, seed=case, perturb=True)
    m = build_model(arch, cfg, sd)
    m.compute_dtype = dtype
    _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=200 + case)
    g = torch.Generator().manual_seed(case)
    shape = (B, J, 1, T)
    tape = torch.randn((steps + 1,) + shape, generator=g)
    y_cpu = {"seed": seedp, "mfcc": mfcc}
    if cfgs:
        y_cpu["scale"] = torch.linspace(0.0, 2.5, B)
    if inpaint:
        mask = torch.zeros(shape, dtype=torch.bool)
        mask[:, : J // 3, :, : T // 2] = True
        y_cpu["inpainting_mask"], y_cpu["inpainted_motion"] = mask, torch.randn(shape, generator=g)
    init = torch.randn(shape, generator=g) if use_init else None
    y = {k: v.to(dev) for k, v in y_cpu.items()}
    model = ClassifierFreeSampleModel(m) if cfgs else m
    df = _diffusion([steps])
    kw = dict(clip_denoised=clip, model_kwargs={"y": y}, progress=False, skip_timesteps=skip, noise_tape=tape.to(dev))
    if init is not None:
        kw["init_image"] = init.to(dev)
    if const:
        kw["const_noise"] = True
    if dump:
        kw["dump_steps"] = dump
    if kind == "ddim":
        kw["eta"] = eta
    out = (df.p_sample_loop if kind == "p" else df.ddim_sample_loop)(model, shape, **kw)
    out = torch.stack(list(out)) if isinstance(out, list) else out

    def model_fn(x, t, yy):
        c = omf.forward(sd, cfg, x, t, yy)
        if not cfgs:
            return c
        u = omf.forward(sd, cfg, x, t, dict(yy, uncond=True))
        return u + yy["scale"].view(-1, 1, 1, 1) * (c - u)
    tab, tmap = osch.make_tables("cosine", 1000, [steps])
    with torch.no_grad():
        want = osamp.sample_loop(model_fn, tab, tmap, shape, tape, y_cpu, kind=kind, eta=eta, skip_timesteps=skip, init_image=init,
                                 const_noise=const, dump_steps=dump, clip_denoised=clip)
    want = torch.stack(list(want)) if isinstance(want, list) else want
    err = rel_err(out.cpu(), want)
    tol = 2e-4 if dtype == "fp32" else 2e-2
    ok = bool(torch.isfinite(out).all()) and err < tol
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} case {case:3d}: {arch:7s} J={J} T={T} B={B} steps={steps} {kind} eta={eta} cfg={int(cfgs)} clip={int(clip)} "
          f"const={int(const)} skip={skip} init={int(use_init)} inpaint={int(inpaint)} dump={dump} {dtype}: rel err {err:.2e} ({time.time() - t00:.0f} s)",
          flush=True)
print(f"{n - bad} / {n} within tolerance")
sys.exit(1 if bad else 0)
