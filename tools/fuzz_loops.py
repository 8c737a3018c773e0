#!/usr/bin/env python3
"""Random-option parity sweep of the fused sampling loop (gdx_sample_loop, noise tape) against the CPU oracle's loop:
python tools/fuzz_loops.py [N] [seed] [--only DTYPE] [--guided].  Draws topology, J, T (token-major and general loop state), B,
schedule length, ancestral / DDIM (eta 0 / 0.5), guidance, clip_denoised, const_noise, skip_timesteps, init_image, inpainting,
dump_steps and the compute dtype; exit code 1 if any case exceeds the tolerance its mode states
(gesturediffusion_amd.numerics.stated_tolerance: fp32 2e-4 here, 16-bit modes 2e-2 of max|ref|, times |s| + |1 - s| under a
guidance scale s).  The oracle is the checker.  draw_cases / run_case are imported by tests/test_gpu_round3.py."""
import os, random, sys, time
_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests"))


def draw_cases(n, seed):
    """The first n option sets of the sweep with this seed (pure host code: the draw order is part of the records in profiles/)."""
    rnd = random.Random(seed)
    out = []
    for case in range(n):
        c = dict(case=case)
        c["arch"] = rnd.choice(["mdm", "mdm_old"])
        c["J"] = rnd.choice([16, 18, 37])
        c["T"] = rnd.choice([20, 30, 40]) if c["arch"] == "mdm" else rnd.choice([12, 18, 20, 33, 36])
        c["B"] = rnd.choice([1, 2, 3])
        c["steps"] = rnd.choice([6, 12])
        c["kind"] = rnd.choice(["p", "p", "ddim"])
        c["eta"] = rnd.choice([0.0, 0.5]) if c["kind"] == "ddim" else 0.0
        c["cfg"], c["clip"] = rnd.random() < 0.4, rnd.random() < 0.4
        c["const"] = c["kind"] == "p" and rnd.random() < 0.25
        c["skip"] = rnd.choice([0, 0, 2])
        c["init"] = rnd.random() < 0.3
        c["inpaint"] = rnd.random() < 0.25
        c["dump"] = [0, c["steps"] - c["skip"] - 1] if c["kind"] == "p" and rnd.random() < 0.2 else None
        c["dtype"] = rnd.choice(["fp32", "fp32", "fp16", "bf16"])
        c["scale_max"] = 2.5 if c["cfg"] and c["B"] > 1 else 0.0          # scale = linspace(0, 2.5, B)
        out.append(c)
    return out


def tolerance(c):
    from gesturediffusion_amd.numerics import stated_tolerance
    if c["dtype"] == "fp32":
        return 2e-4                                                      # the suite's own, tighter than the stated 1e-3
    return stated_tolerance(c["dtype"], c["scale_max"] if c["cfg"] else None)


def run_case(c, scale=None):
    """One loop on the GPU against the oracle's loop on the same tape; returns err / max|ref|.  scale: guidance vector override."""
    import torch
    from test_gpu_parity import build_model, rel_err, _diffusion
    from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs
    from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from oracle import mdm_forward as omf, sampler as osamp, schedule as osch
    dev = torch.device("cuda:0")
    arch, J, T, B, steps, kind, eta, case = c["arch"], c["J"], c["T"], c["B"], c["steps"], c["kind"], c["eta"], c["case"]
    cfg = dict(arch=arch, njoints=J, nfeats=1, latent_dim=128, ff_size=256, num_layers=2, num_heads=4, seed_poses=10)
    sd = init_state_dict(cfg, seed=case, perturb=True)
    m = build_model(arch, cfg, sd)
    m.compute_dtype = c["dtype"]
    _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=200 + case)
    g = torch.Generator().manual_seed(case)
    shape = (B, J, 1, T)
    tape = torch.randn((steps + 1,) + shape, generator=g)
    y_cpu = {"seed": seedp, "mfcc": mfcc}
    if c["cfg"]:
        y_cpu["scale"] = torch.linspace(0.0, 2.5, B) if scale is None else torch.as_tensor(scale, dtype=torch.float32).expand(B).clone()
    if c["inpaint"]:
        mask = torch.zeros(shape, dtype=torch.bool)
        mask[:, : J // 3, :, : T // 2] = True
        y_cpu["inpainting_mask"], y_cpu["inpainted_motion"] = mask, torch.randn(shape, generator=g)
    init = torch.randn(shape, generator=g) if c["init"] else None
    y = {k: v.to(dev) for k, v in y_cpu.items()}
    model = ClassifierFreeSampleModel(m) if c["cfg"] else m
    df = _diffusion([steps])
    kw = dict(clip_denoised=c["clip"], model_kwargs={"y": y}, progress=False, skip_timesteps=c["skip"], noise_tape=tape.to(dev))
    if init is not None:
        kw["init_image"] = init.to(dev)
    if c["const"]:
        kw["const_noise"] = True
    if c["dump"]:
        kw["dump_steps"] = c["dump"]
    if kind == "ddim":
        kw["eta"] = eta
    out = (df.p_sample_loop if kind == "p" else df.ddim_sample_loop)(model, shape, **kw)
    out = torch.stack(list(out)) if isinstance(out, list) else out

    def model_fn(x, t, yy):
        cc = omf.forward(sd, cfg, x, t, yy)
        if not c["cfg"]:
            return cc
        u = omf.forward(sd, cfg, x, t, dict(yy, uncond=True))
        return u + yy["scale"].view(-1, 1, 1, 1) * (cc - u)
    tab, tmap = osch.make_tables("cosine", 1000, [steps])
    with torch.no_grad():
        want = osamp.sample_loop(model_fn, tab, tmap, shape, tape, y_cpu, kind=kind, eta=eta, skip_timesteps=c["skip"], init_image=init,
                                 const_noise=c["const"], dump_steps=c["dump"], clip_denoised=c["clip"])
    want = torch.stack(list(want)) if isinstance(want, list) else want
    if not bool(torch.isfinite(out).all()):
        return float("inf")
    return rel_err(out.cpu(), want)


def describe(c):
    return (f"case {c['case']:3d}: {c['arch']:7s} J={c['J']} T={c['T']} B={c['B']} steps={c['steps']} {c['kind']} eta={c['eta']} "
            f"cfg={int(c['cfg'])} clip={int(c['clip'])} const={int(c['const'])} skip={c['skip']} init={int(c['init'])} "
            f"inpaint={int(c['inpaint'])} dump={c['dump']} {c['dtype']}")


def main(argv):
    pos = [a for a in argv if not a.startswith("--")]
    n = int(pos[0]) if pos else 40
    seed = int(pos[1]) if len(pos) > 1 else 1
    only = argv[argv.index("--only") + 1] if "--only" in argv else None
    guided = "--guided" in argv
    bad, ran, t00 = 0, 0, time.time()
    for c in draw_cases(n, seed):
        if (only and c["dtype"] != only) or (guided and not c["cfg"]):
            continue
        err, tol = run_case(c), tolerance(c)
        ok = err < tol
        bad += not ok
        ran += 1
        print(f"{'ok ' if ok else 'BAD'} {describe(c)}: rel err {err:.2e} (tol {tol:.1e}; {time.time() - t00:.0f} s)", flush=True)
    print(f"{ran - bad} / {ran} within tolerance")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
