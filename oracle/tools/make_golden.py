#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Test infrastructure.  Imports the reference's Python modules from /root/reference
(never copied), with the three harness-side shims of SURVEY.md section 8c that do not
touch the path's arithmetic:
  1. np.float / np.int aliases (removed in NumPy >= 1.24, used at import time by
     data_loaders/humanml/common/quaternion.py:13 via diffusion/gaussian_diffusion.py:18);
  2. empty stub modules for `clip`, `smplx`, `smplx.lbs` (imported at model/mdm.py:5 and
     model/smpl.py:7-8, never called with use_text=False);
  3. model.mdm.Rotation2xyz / model.mdm_old.Rotation2xyz replaced by a stub with
     `.smpl_model = nn.Identity()` (the real ctor reads ./body_models/smpl/*.pkl).

Fixtures hold DATA only: inputs, weights of tiny models, expected outputs/intermediates.
Real-shape cases store outputs only; weights and inputs are regenerated on both sides
from gesturediffusion_amd.utils.init (deterministic).

Usage:  python oracle/tools/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)

from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs  # noqa: E402


def import_reference(ref):
    np.float = float   # shim 1
    np.int = int
    for name in ("clip", "smplx", "smplx.lbs"):  # shim 2
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["smplx"].SMPLLayer = object
    sys.modules["smplx.lbs"].vertices2joints = None
    sys.path.insert(0, ref)
    with contextlib.redirect_stdout(io.StringIO()):
        import model.mdm as ref_mdm
        import model.mdm_old as ref_old
        import model.cfg_sampler as ref_cfg
        import diffusion.gaussian_diffusion as gd
        import diffusion.respace as rs

    class _Rot:  # shim 3
        def __init__(self, *a, **k):
            self.smpl_model = torch.nn.Identity()

        def __call__(self, x, **k):
            return x

    ref_mdm.Rotation2xyz = _Rot
    ref_old.Rotation2xyz = _Rot
    return ref_mdm, ref_old, ref_cfg, gd, rs


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build_ref_model(mods, cfg, sd, double=False):
    ref_mdm, ref_old = mods[0], mods[1]
    kw = dict(njoints=cfg["njoints"], nfeats=cfg["nfeats"], translation=True, pose_rep="rot6d", glob=True,
              glob_rot=True, latent_dim=cfg["latent_dim"], ff_size=cfg["ff_size"], num_layers=cfg["num_layers"],
              num_heads=cfg["num_heads"], dropout=0.1, activation="gelu", data_rep="genea_vec",
              cond_mask_prob=0.1, dataset="genea2023", use_text=False, mfcc_input=True, use_wav_enc=False,
              seed_poses=cfg["seed_poses"], use_audio=False, modeltype="", clip_version="ViT-B/32")
    cls = ref_mdm.MDM if cfg["arch"] == "mdm" else ref_old.MDM_Old
    m = quiet(cls, **kw)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(".pe") or k.endswith("inv_freq") for k in missing), missing
    m.eval()
    if double:
        m.double()
    return m


def make_diffusion(gd, rs, respacing, schedule="cosine", steps=1000, mean_type="START_X", var_type="FIXED_SMALL"):
    betas = gd.get_named_beta_schedule(schedule, steps, 1.0)
    return rs.SpacedDiffusion(
        use_timesteps=rs.space_timesteps(steps, respacing if respacing else [steps]), betas=betas,
        model_mean_type=getattr(gd.ModelMeanType, mean_type), model_var_type=getattr(gd.ModelVarType, var_type),
        loss_type=gd.LossType.MSE, rescale_timesteps=False)


TABLE_NAMES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
               "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
               "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
               "posterior_mean_coef1", "posterior_mean_coef2"]

TINY = dict(njoints=16, nfeats=1, latent_dim=128, ff_size=256, num_layers=2, num_heads=4, seed_poses=10)


def tiny_cfg(arch):
    return dict(TINY, arch=arch)


def real_cfg(arch, J, d, L=8):
    return dict(arch=arch, njoints=J, nfeats=1, latent_dim=d, ff_size=1024, num_layers=L, num_heads=4, seed_poses=10)


def gen_schedule(mods, out):
    gd, rs = mods[3], mods[4]
    d = {}
    for sched in ("cosine", "linear"):
        for tag, resp in (("1000", ""), ("ddim10", "ddim10"), ("ddim100", "ddim100"), ("s10", [10]),
                          ("s100", [100]), ("s20", [20])):
            df = make_diffusion(gd, rs, resp, sched)
            for n in TABLE_NAMES:
                d[f"{sched}.{tag}.{n}"] = getattr(df, n)
            d[f"{sched}.{tag}.timestep_map"] = np.array(df.timestep_map, dtype=np.int64)
    np.savez_compressed(os.path.join(out, "schedule.npz"), **d)


def capture_taps(model, arch):
    taps = {}
    hooks = []

    def save(name):
        def hook(_m, _i, o):
            taps[name] = o.detach().clone()
        return hook

    hooks.append(model.input_process.register_forward_hook(save("emb_pose" if arch == "mdm" else "input_linear")))
    if arch == "mdm":
        hooks.append(model.project_to_lat.register_forward_hook(save("project_to_lat")))
        hooks.append(model.cross_local_attention.register_forward_hook(save("local_attn_raw")))
        hooks.append(model.cross_local_attention.register_forward_pre_hook(
            lambda _m, i: taps.__setitem__("rope1", i[0].detach().clone())))
    hooks.append(model.seqTransEncoder.register_forward_pre_hook(
        lambda _m, i: taps.__setitem__("enc_in", i[0].detach().clone())))
    for l, layer in enumerate(model.seqTransEncoder.layers):
        hooks.append(layer.register_forward_hook(save(f"seqTransEncoder.layers.{l}.out")))
    return taps, hooks


def gen_forward_tiny(mods, out):
    for arch in ("mdm", "mdm_old"):
        cfg = tiny_cfg(arch)
        sd = init_state_dict(cfg, seed=1, perturb=True)
        B, T = 2, 20
        x, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=3)
        t = torch.tensor([7, 993])
        m = build_ref_model(mods, cfg, sd)
        d = {"w." + k: v.numpy() for k, v in sd.items()}
        d.update(x=x.numpy(), t=t.numpy(), seed=seedp.numpy(), mfcc=mfcc.numpy())
        for uncond in (False, True):
            taps, hooks = capture_taps(m, arch)
            y = {"seed": seedp, "mfcc": mfcc}
            if uncond:
                y["uncond"] = True
            with torch.no_grad():
                o = m(x, t, y)
            for h in hooks:
                h.remove()
            tag = "uncond." if uncond else "cond."
            d[tag + "out"] = o.contiguous().numpy()
            for k, v in taps.items():
                d[tag + "tap." + k] = v.contiguous().numpy()
        # F5: fp64 run of the same model -> fp32 noise floor calibration
        m64 = build_ref_model(mods, cfg, sd, double=True)
        with torch.no_grad():
            o64 = m64(x.double(), t, {"seed": seedp.double(), "mfcc": mfcc.double()})
        d["cond.out_fp64"] = o64.contiguous().numpy()
        np.savez_compressed(os.path.join(out, f"forward_{arch}_tiny.npz"), **d)


class TapeNoise:
    """Replays a pre-drawn noise tape through torch.randn_like (harness-side)."""

    def __init__(self, tape):
        self.tape, self.k = tape, 0

    def __enter__(self):
        self._orig = torch.randn_like

        def fake(x, *a, **k):
            z = self.tape[self.k]
            self.k += 1
            assert z.shape == x.shape
            return z.to(x.dtype)
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self._orig


def gen_loops_tiny(mods, out):
    ref_cfg, gd, rs = mods[2], mods[3], mods[4]
    for arch in ("mdm", "mdm_old"):
        cfg = tiny_cfg(arch)
        sd = init_state_dict(cfg, seed=2, perturb=True)
        B, T = 3, 20
        _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=5)
        m = build_ref_model(mods, cfg, sd)
        cfgm = ref_cfg.ClassifierFreeSampleModel(m)
        g = torch.Generator().manual_seed(1234)
        shape = (B, cfg["njoints"], 1, T)
        tape = torch.randn(22, *shape, generator=g)
        init_image = torch.randn(*shape, generator=g)
        mask = torch.zeros(shape, dtype=torch.bool)
        mask[..., :5] = True
        mask[:, :4] = True
        motion = torch.randn(*shape, generator=g)
        scale = torch.tensor([2.5, 1.0, 0.0])
        d = {"w." + k: v.numpy() for k, v in sd.items()}
        d.update(seed=seedp.numpy(), mfcc=mfcc.numpy(), tape=tape.numpy(), init_image=init_image.numpy(),
                 inpainting_mask=mask.numpy(), inpainted_motion=motion.numpy(), scale=scale.numpy())

        def run(kind, respacing, model, y, **kw):
            df = make_diffusion(gd, rs, respacing)
            fn = df.p_sample_loop if kind == "p" else df.ddim_sample_loop
            with TapeNoise(tape[1:]):
                r = fn(model, shape, noise=tape[0].clone(), clip_denoised=False, model_kwargs={"y": y},
                       progress=False, **kw)
            return r

        y = {"seed": seedp, "mfcc": mfcc}
        ycfg = dict(y, scale=scale)
        d["p20"] = run("p", [20], m, y).numpy()
        d["p20_cfg"] = run("p", [20], cfgm, ycfg).numpy()
        d["ddim10"] = run("ddim", "ddim10", m, y).numpy()
        d["ddim10_cfg"] = run("ddim", "ddim10", cfgm, ycfg).numpy()
        d["ddim10_eta05"] = run("ddim", "ddim10", m, y, eta=0.5).numpy()
        d["p20_const_noise"] = run("p", [20], m, y, const_noise=True).numpy()
        dumped = run("p", [20], m, y, dump_steps=[0, 9, 19])
        d["p20_dump"] = torch.stack(dumped).numpy()
        d["p20_init_skip"] = run("p", [20], m, y, init_image=init_image, skip_timesteps=5).numpy()
        d["p20_skip_only"] = run("p", [20], m, y, skip_timesteps=8).numpy()
        yin = dict(y, inpainting_mask=mask, inpainted_motion=motion)
        d["p20_inpaint"] = run("p", [20], m, yin).numpy()
        d["p20_cfg_inpaint"] = run("p", [20], cfgm, dict(yin, scale=scale)).numpy()
        # ddim_reverse_sample (gaussian_diffusion.py:841-877): three deterministic steps x_t -> x_{t+1}
        df = make_diffusion(gd, rs, "ddim10")
        xr = tape[0].clone()
        with torch.no_grad():
            for ti in (0, 1, 2):
                xr = df.ddim_reverse_sample(m, xr, torch.tensor([ti] * B), clip_denoised=False,
                                            model_kwargs={"y": y})["sample"]
        d["ddim10_reverse3"] = xr.numpy()
        np.savez_compressed(os.path.join(out, f"loops_{arch}_tiny.npz"), **d)


def gen_plms_tiny(mods, out):
    """plms_sample_loop (gaussian_diffusion.py:995-1190) on the same tiny models / inputs as gen_loops_tiny
    (outputs only; weights and inputs are read from loops_{arch}_tiny.npz by the tests)."""
    ref_cfg, gd, rs = mods[2], mods[3], mods[4]
    for arch in ("mdm", "mdm_old"):
        cfg = tiny_cfg(arch)
        sd = init_state_dict(cfg, seed=2, perturb=True)
        B, T = 3, 20
        _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=5)
        m = build_ref_model(mods, cfg, sd)
        cfgm = ref_cfg.ClassifierFreeSampleModel(m)
        g = torch.Generator().manual_seed(1234)
        shape = (B, cfg["njoints"], 1, T)
        tape = torch.randn(22, *shape, generator=g)
        init_image = torch.randn(*shape, generator=g)
        mask = torch.zeros(shape, dtype=torch.bool)
        mask[..., :5] = True
        mask[:, :4] = True
        motion = torch.randn(*shape, generator=g)
        scale = torch.tensor([2.5, 1.0, 0.0])
        y = {"seed": seedp, "mfcc": mfcc}
        d = {}

        def run(model, yy, **kw):
            df = make_diffusion(gd, rs, [10])
            return df.plms_sample_loop(model, shape, noise=tape[0].clone(), clip_denoised=False, model_kwargs={"y": yy},
                                       progress=False, **kw).numpy()
        d["plms10_o2"] = run(m, y)
        d["plms10_o3"] = run(m, y, order=3)
        d["plms10_o4_cfg"] = run(cfgm, dict(y, scale=scale), order=4)
        d["plms10_o2_inpaint"] = run(m, dict(y, inpainting_mask=mask, inpainted_motion=motion))
        d["plms10_o2_init_skip"] = run(m, y, init_image=init_image, skip_timesteps=3)
        try:
            run(m, y, order=1)
            d["plms_order1_error"] = np.array("none")
        except Exception as e:  # noqa: BLE001
            d["plms_order1_error"] = np.array(type(e).__name__)
        np.savez_compressed(os.path.join(out, f"plms_{arch}_tiny.npz"), **d)


def denoised_fn_fixture(x):
    """Deterministic stand-in for a user's denoised_fn (process_xstart, gaussian_diffusion.py:349-355)."""
    return 1.5 * torch.tanh(x) + 0.05


def gen_clip_tiny(mods, out):
    """p_sample_loop / ddim_sample_loop with clip_denoised=True and with a denoised_fn (process_xstart :349-355: the
    inpainting blend :307-311 comes first, then denoised_fn, then the clamp), same tiny models / inputs / noise tape as
    loops_{arch}_tiny.npz (outputs only)."""
    ref_cfg, gd, rs = mods[2], mods[3], mods[4]
    for arch in ("mdm", "mdm_old"):
        cfg = tiny_cfg(arch)
        sd = init_state_dict(cfg, seed=2, perturb=True)
        B, T = 3, 20
        _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=5)
        m = build_ref_model(mods, cfg, sd)
        cfgm = ref_cfg.ClassifierFreeSampleModel(m)
        g = torch.Generator().manual_seed(1234)
        shape = (B, cfg["njoints"], 1, T)
        tape = torch.randn(22, *shape, generator=g)
        torch.randn(*shape, generator=g)                      # (init_image of the loops fixture: keeps the generator in step)
        mask = torch.zeros(shape, dtype=torch.bool)
        mask[..., :5] = True
        mask[:, :4] = True
        motion = torch.randn(*shape, generator=g) * 2.0       # beyond [-1, 1]: the clamp acts on the inpainted values too
        scale = torch.tensor([2.5, 1.0, 0.0])
        y = {"seed": seedp, "mfcc": mfcc}
        d = {"inpainted_motion": motion.numpy()}

        def run(kind, respacing, model, yy, **kw):
            df = make_diffusion(gd, rs, respacing)
            fn = df.p_sample_loop if kind == "p" else df.ddim_sample_loop
            with TapeNoise(tape[1:]):
                return fn(model, shape, noise=tape[0].clone(), model_kwargs={"y": yy}, progress=False, **kw).numpy()
        yin = dict(y, inpainting_mask=mask, inpainted_motion=motion)
        d["p20_clip"] = run("p", [20], m, y, clip_denoised=True)
        d["p20_clip_cfg_inpaint"] = run("p", [20], cfgm, dict(yin, scale=scale), clip_denoised=True)
        d["ddim10_clip"] = run("ddim", "ddim10", m, y, clip_denoised=True)
        d["p20_dfn_inpaint"] = run("p", [20], m, yin, clip_denoised=False, denoised_fn=denoised_fn_fixture)
        d["p20_dfn_clip"] = run("p", [20], m, y, clip_denoised=True, denoised_fn=denoised_fn_fixture)
        np.savez_compressed(os.path.join(out, f"clip_{arch}_tiny.npz"), **d)


def gen_meantypes_tiny(mods, out):
    """The mean / variance parametrisations of p_mean_variance other than the configured START_X + FIXED_SMALL
    (:316-372): the denoiser's output read as EPSILON (_predict_xstart_from_eps :390-396) or as PREVIOUS_X
    (_predict_xstart_from_xprev :398-405, mean = the output itself), and FIXED_LARGE variance -- whole loops of the
    reference on the tiny models, same inputs / noise tape as loops_{arch}_tiny.npz (outputs only)."""
    ref_cfg, gd, rs = mods[2], mods[3], mods[4]
    for arch in ("mdm", "mdm_old"):
        cfg = tiny_cfg(arch)
        sd = init_state_dict(cfg, seed=2, perturb=True)
        B, T = 3, 20
        _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=5)
        m = build_ref_model(mods, cfg, sd)
        cfgm = ref_cfg.ClassifierFreeSampleModel(m)
        g = torch.Generator().manual_seed(1234)
        shape = (B, cfg["njoints"], 1, T)
        tape = torch.randn(22, *shape, generator=g)
        torch.randn(*shape, generator=g)                      # (init_image of the loops fixture)
        mask = torch.zeros(shape, dtype=torch.bool)
        mask[..., :5] = True
        mask[:, :4] = True
        motion = torch.randn(*shape, generator=g)
        scale = torch.tensor([2.5, 1.0, 0.0])
        y = {"seed": seedp, "mfcc": mfcc}
        yin = dict(y, inpainting_mask=mask, inpainted_motion=motion, scale=scale)
        d = {}

        def run(kind, respacing, model, yy, mean_type, var_type="FIXED_SMALL", **kw):
            df = make_diffusion(gd, rs, respacing, mean_type=mean_type, var_type=var_type)
            fn = {"p": df.p_sample_loop, "ddim": df.ddim_sample_loop, "plms": df.plms_sample_loop}[kind]
            with TapeNoise(tape[1:]):
                return fn(model, shape, noise=tape[0].clone(), model_kwargs={"y": yy}, progress=False, **kw).numpy()
        for tag, mt in (("eps", "EPSILON"), ("prevx", "PREVIOUS_X")):
            d[f"{tag}_p20"] = run("p", [20], m, y, mt, clip_denoised=False)
            d[f"{tag}_p20_clip"] = run("p", [20], m, y, mt, clip_denoised=True)
            d[f"{tag}_p20_clip_cfg"] = run("p", [20], cfgm, dict(y, scale=scale), mt, clip_denoised=True)
            try:                                                  # the inpainting blend asserts START_X (:309)
                run("p", [20], m, yin, mt, clip_denoised=True)
                raise SystemExit("expected the reference to refuse inpainting with " + mt)
            except AssertionError:
                pass
            d[f"{tag}_p20_large"] = run("p", [20], m, y, mt, var_type="FIXED_LARGE", clip_denoised=True)
            d[f"{tag}_ddim10_clip"] = run("ddim", "ddim10", m, y, mt, clip_denoised=True)
            d[f"{tag}_ddim10_eta05"] = run("ddim", "ddim10", m, y, mt, clip_denoised=True, eta=0.5)
            d[f"{tag}_p20_dfn"] = run("p", [20], m, y, mt, clip_denoised=True, denoised_fn=denoised_fn_fixture)
            d[f"{tag}_plms10_clip"] = run("plms", "ddim10", m, y, mt, clip_denoised=True, order=2)
        d["startx_p20_large"] = run("p", [20], m, y, "START_X", var_type="FIXED_LARGE", clip_denoised=False)
        for k, v in d.items():
            assert np.isfinite(v).all(), (arch, k)
        np.savez_compressed(os.path.join(out, f"meantypes_{arch}_tiny.npz"), **d)


def cond_fn_fixture(x, t, **kwargs):
    """Deterministic stand-in for a classifier gradient (the reference has no classifier): smooth in x, depends on t."""
    return 0.05 * torch.sin(x) * (1.0 + t.view(-1, 1, 1, 1).float() / 1000.0)


def gen_guided_tiny(mods, out):
    """p_sample_loop / ddim_sample_loop with cond_fn (condition_mean / condition_score, :418-494) on the tiny models,
    same inputs and noise tape as loops_{arch}_tiny.npz (outputs only)."""
    gd, rs = mods[3], mods[4]
    for arch in ("mdm", "mdm_old"):
        cfg = tiny_cfg(arch)
        sd = init_state_dict(cfg, seed=2, perturb=True)
        B, T = 3, 20
        _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=5)
        m = build_ref_model(mods, cfg, sd)
        g = torch.Generator().manual_seed(1234)
        shape = (B, cfg["njoints"], 1, T)
        tape = torch.randn(22, *shape, generator=g)
        y = {"seed": seedp, "mfcc": mfcc}
        d = {}
        for name, kind, resp, kw in (("p20_guided", "p", [20], {}), ("ddim10_guided", "ddim", "ddim10", {}),
                                     ("ddim10_eta05_guided", "ddim", "ddim10", {"eta": 0.5})):
            df = make_diffusion(gd, rs, resp)
            fn = df.p_sample_loop if kind == "p" else df.ddim_sample_loop
            with TapeNoise(tape[1:]):
                d[name] = fn(m, shape, noise=tape[0].clone(), clip_denoised=False, model_kwargs={"y": y}, progress=False,
                             cond_fn=cond_fn_fixture, **kw).numpy()
        df = make_diffusion(gd, rs, [10])
        d["plms10_o2_guided"] = df.plms_sample_loop(m, shape, noise=tape[0].clone(), clip_denoised=False, model_kwargs={"y": y},
                                                    progress=False, cond_fn=cond_fn_fixture).numpy()
        np.savez_compressed(os.path.join(out, f"guided_{arch}_tiny.npz"), **d)


def helper_inputs(n_steps):
    """Inputs of the public helper methods: pose-shaped tensors, per-sample timesteps incl. 0 and the last step."""
    g = torch.Generator().manual_seed(77)
    shape = (4, 16, 1, 20)
    x_start, x_t, pred = (torch.randn(shape, generator=g) for _ in range(3))
    t = torch.tensor([0, 1, n_steps // 2, n_steps - 1])
    return x_start, x_t, pred, t


def gen_helpers(mods, out):
    """q_mean_variance (:216), q_posterior_mean_variance (:253), condition_mean (:418), condition_score (:448) of the reference's
    GaussianDiffusion / SpacedDiffusion, on the full schedule and on a 20-step respacing (cond_fn then sees mapped timesteps)."""
    gd, rs = mods[3], mods[4]
    d = {}
    for tag, resp in (("full", ""), ("r20", [20])):
        df = make_diffusion(gd, rs, resp)
        x_start, x_t, pred, t = helper_inputs(df.num_timesteps)
        qm = df.q_mean_variance(x_start, t)
        qp = df.q_posterior_mean_variance(x_start, x_t, t)
        for i, nm in enumerate(("mean", "variance", "log_variance")):
            d[f"{tag}_qmv_{nm}"] = qm[i].numpy().copy()
            d[f"{tag}_qpost_{nm}"] = qp[i].numpy().copy()
        pmv = {"mean": qp[0], "variance": qp[1], "log_variance": qp[2], "pred_xstart": pred}
        # the gradient itself is part of the fixture: torch.sin is not bit-reproducible across CPUs (vectorised libm variants),
        # the four methods under test only consume it
        d[f"{tag}_grad"] = cond_fn_fixture(x_t, torch.tensor(df.timestep_map)[t]).numpy().copy()
        d[f"{tag}_condition_mean"] = df.condition_mean(cond_fn_fixture, pmv, x_t, t, model_kwargs={}).numpy().copy()
        cs = df.condition_score(cond_fn_fixture, pmv, x_t, t, model_kwargs={})
        d[f"{tag}_condition_score_pred_xstart"] = cs["pred_xstart"].numpy().copy()
        d[f"{tag}_condition_score_mean"] = cs["mean"].numpy().copy()
        assert cs["variance"] is pmv["variance"]
    np.savez_compressed(os.path.join(out, "helpers.npz"), **d)


def gen_losses_tiny(mods, out):
    """training_losses (gaussian_diffusion.py:1227-1352), forward values only, on the tiny models: per-sample timesteps,
    explicit noise, a ragged frame mask.  The reference reads `model.model` (its DDP / CFG wrapper convention)."""
    gd, rs = mods[3], mods[4]
    for arch in ("mdm", "mdm_old"):
        cfg = tiny_cfg(arch)
        sd = init_state_dict(cfg, seed=2, perturb=True)
        B, T = 3, 20
        _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=5)
        m = build_ref_model(mods, cfg, sd)

        class Wrapped(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.model = inner

            def forward(self, x, t, **kw):
                return self.model(x, t, **kw)
        g = torch.Generator().manual_seed(99)
        shape = (B, cfg["njoints"], 1, T)
        x_start = torch.randn(*shape, generator=g)
        noise = torch.randn(*shape, generator=g)
        mask = torch.ones(B, 1, 1, T, dtype=torch.bool)
        mask[1, ..., 13:] = False
        mask[2, ..., 7:] = False
        d = {"x_start": x_start.numpy(), "noise": noise.numpy(), "mask": mask.numpy()}
        for tag, resp, t in (("full", [1000], torch.tensor([0, 500, 999])), ("r20", [20], torch.tensor([19, 3, 0]))):
            df = make_diffusion(gd, rs, resp)
            with torch.no_grad():
                terms = df.training_losses(Wrapped(m), x_start, t, model_kwargs={"y": {"seed": seedp, "mfcc": mfcc, "mask": mask}},
                                           noise=noise)
            d[tag + ".t"] = t.numpy()
            for k, v in terms.items():
                d[f"{tag}.{k}"] = v.numpy()
        np.savez_compressed(os.path.join(out, f"losses_{arch}_tiny.npz"), **d)


CHUNKS = dict(njoints=18, n_chunks=3, B=3, T=20, respacing=[20], scale=2.5, tape_seed=4321)


def chunk_inputs(cfg=None):
    """Inputs of the chunked-driver fixture, regenerated identically by the generator and the tests: weights, first seed
    poses, per-chunk MFCCs and per-chunk noise tapes (x_T + one draw per step)."""
    c = CHUNKS
    cfg = cfg or dict(TINY, arch="mdm", njoints=c["njoints"])
    sd = init_state_dict(cfg, seed=6, perturb=True)
    _, seedp, _ = synthetic_inputs(cfg, c["B"], c["T"], seed=8)
    g = torch.Generator().manual_seed(c["tape_seed"])
    mfccs = [torch.randn(c["B"], 26, 1, c["T"], generator=g) for _ in range(c["n_chunks"])]
    tapes = [torch.randn(c["respacing"][0] + 1, c["B"], c["njoints"], 1, c["T"], generator=g) for _ in range(c["n_chunks"])]
    return cfg, sd, seedp, mfccs, tapes


def gen_chunks_tiny(mods, out):
    """the reference's chunk loop (`sample/generate.py:91-130`) is three statements around its own
    `p_sample_loop` -- build y for the chunk, replace y['seed'] by `sample_out[..., -seed_poses:]` after the first chunk
    (`:104-107`), add the guidance scale (`:114-115`).  `sample/generate.py` itself cannot be imported (bvhsdk, the GENEA
    dataset), so those three statements are restated here, harness-side, around the REFERENCE's ClassifierFreeSampleModel
    and p_sample_loop; the outputs pin the build's `sample_chunks` driver (seed hand-off through a non-contiguous view,
    conditioning re-encoded per chunk)."""
    ref_cfg, gd, rs = mods[2], mods[3], mods[4]
    c = CHUNKS
    d = {}
    for arch in ("mdm", "mdm_old"):
        cfg, sd, seedp, mfccs, tapes = chunk_inputs(dict(TINY, arch=arch, njoints=c["njoints"]))
        model = ref_cfg.ClassifierFreeSampleModel(build_ref_model(mods, cfg, sd))
        df = make_diffusion(gd, rs, c["respacing"])
        shape = (c["B"], c["njoints"], 1, c["T"])
        sample_out = None
        for chunk in range(c["n_chunks"]):
            y = {"mfcc": mfccs[chunk], "seed": seedp}
            if chunk > 0:
                y["seed"] = sample_out[..., -cfg["seed_poses"]:]
            y["scale"] = torch.ones(c["B"]) * c["scale"]
            with TapeNoise(tapes[chunk][1:]):
                sample_out = df.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0,
                                              init_image=None, progress=False, dump_steps=None,
                                              noise=tapes[chunk][0].clone(), const_noise=False)
            d[f"{arch}.chunk{chunk}"] = sample_out.numpy()
    np.savez_compressed(os.path.join(out, "chunks_tiny.npz"), **d)


def gen_real_shapes(mods, out):
    """F4: outputs only; weights/inputs regenerate from gesturediffusion_amd.utils.init."""
    cases = {
        "c1_v2": (real_cfg("mdm", 150, 512), 4, 60),
        "c2_v1": (real_cfg("mdm_old", 263, 512), 2, 196),
        "c2_v2": (real_cfg("mdm", 263, 512), 2, 200),
        "c5_v2": (real_cfg("mdm", 498, 1024), 1, 520),
    }
    d = {}
    for name, (cfg, B, T) in cases.items():
        sd = init_state_dict(cfg, seed=0)
        x, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=10)
        t = torch.arange(B) * 37 + 500
        m = build_ref_model(mods, cfg, sd)
        with torch.no_grad():
            o = m(x, t, {"seed": seedp, "mfcc": mfcc})
            ou = m(x, t, {"seed": seedp, "mfcc": mfcc, "uncond": True})
        d[name + ".out"] = o.contiguous().numpy()
        d[name + ".out_uncond"] = ou.contiguous().numpy()
        d[name + ".t"] = t.numpy()
        d[name + ".meta"] = np.array([B, T, cfg["njoints"], cfg["latent_dim"]])
    # C1 end to end: 10-step DDIM, B=4, T=60 with a torch-generated tape (stored as seed only)
    cfg, B, T = cases["c1_v2"]
    sd = init_state_dict(cfg, seed=0)
    _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=10)
    m = build_ref_model(mods, cfg, sd)
    g = torch.Generator().manual_seed(77)
    tape = torch.randn(11, B, cfg["njoints"], 1, T, generator=g)
    df = make_diffusion(mods[3], mods[4], "ddim10")
    with TapeNoise(tape[1:]):
        r = df.ddim_sample_loop(m, (B, cfg["njoints"], 1, T), noise=tape[0].clone(), clip_denoised=False,
                                model_kwargs={"y": {"seed": seedp, "mfcc": mfcc}}, progress=False)
    d["c1_v2.ddim10"] = r.numpy()
    np.savez_compressed(os.path.join(out, "real_shapes.npz"), **d)


def gen_negative(mods, out):
    """F6: the reference's failure modes, recorded as exception class names."""
    ref_mdm = mods[0]
    cfg = tiny_cfg("mdm")
    sd = init_state_dict(cfg, seed=1)
    m = build_ref_model(mods, cfg, sd)
    res = {}

    def exc_name(fn):
        try:
            with torch.no_grad():
                quiet(fn)
            return "none"
        except Exception as e:  # noqa: BLE001
            return type(e).__name__

    x, seedp, mfcc = synthetic_inputs(cfg, 2, 16, seed=3)     # T=16 not a multiple of 10
    t = torch.tensor([1, 2])
    res["v2_T_not_multiple_of_10"] = exc_name(lambda: m(x, t, {"seed": seedp, "mfcc": mfcc}))
    x, seedp, mfcc = synthetic_inputs(cfg, 2, 20, seed=3)
    res["missing_seed"] = exc_name(lambda: m(x, t, {"mfcc": mfcc}))
    res["missing_mfcc_key"] = exc_name(lambda: m(x, t, {"seed": seedp}))

    def bad_rep():
        ref_mdm.InputProcess("rot6d", 16, 128)(x)
    res["data_rep_not_genea_vec"] = exc_name(bad_rep)
    res["eval_returns_none"] = str(m.eval() is None)
    with open(os.path.join(out, "negative_cases.txt"), "w") as f:
        for k, v in res.items():
            f.write(f"{k}={v}\n")


def collate_inputs(seed=5):
    """Ragged GENEA-style items (`data_loaders/gesture/data/dataset.py` __getitem__ order: motion [len, J], text,
    length, audio [len * 735], mfcc [len, 26], seed poses [n_seed, J]); shared by the generator and the tests."""
    rng = np.random.default_rng(seed)
    items = []
    for n in (12, 7, 12, 9):
        items.append((rng.normal(size=(n, 6)), f"take{n}", n, rng.normal(size=n * 5).astype(np.float32),
                      rng.normal(size=(n, 26)), rng.normal(size=(4, 6))))
    return items


def gen_collate(mods, out):
    """`gg_collate` / `collate` / `collate_tensors` / `lengths_to_mask` of `data_loaders/tensors.py:3-66`."""
    import importlib
    ten = importlib.import_module("data_loaders.tensors")
    items = collate_inputs()
    # audio and mfcc are concatenated along dim 0 (`tensors.py:43-49`), so their trailing sizes must agree: crop to the shortest
    nmin = min(x[2] for x in items)
    items = [(x[0], x[1], x[2], x[3][:nmin * 5], x[4][:nmin], x[5]) for x in items]
    motion, cond = ten.gg_collate(items)
    y = cond["y"]
    ragged = ten.collate_tensors([torch.ones(2, 3), torch.ones(1, 5) * 2, torch.ones(3, 1) * 3])
    np.savez(os.path.join(out, "collate.npz"), motion=motion.numpy(), mask=y["mask"].numpy(), lengths=y["lengths"].numpy(),
             mfcc=y["mfcc"].numpy(), audio=y["audio"].numpy(), seed=y["seed"].numpy(), text=np.array(y["text"]),
             ragged=ragged.numpy(), len_mask=ten.lengths_to_mask(torch.tensor([0, 3, 5]), 5).numpy())


def gen_state_dict_keys(mods, out):
    """State-dict key -> shape listing of the reference modules (drop-in contract)."""
    with open(os.path.join(out, "state_dict_keys.txt"), "w") as f:
        for arch, J, d in (("mdm", 263, 512), ("mdm_old", 263, 512), ("mdm", 498, 256)):
            cfg = real_cfg(arch, J, d, L=2)
            m = build_ref_model(mods, cfg, init_state_dict(cfg, seed=0))
            for k, v in m.state_dict().items():
                f.write(f"{arch} {J} {d} {k} {'x'.join(str(i) for i in v.shape)}\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="", help="comma-separated subset: schedule,forward,loops,plms,losses,guided,helpers,clip,meantypes,collate,chunks,real,negative,keys")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    mods = import_reference(args.ref)
    gens = {"schedule": gen_schedule, "forward": gen_forward_tiny, "loops": gen_loops_tiny, "plms": gen_plms_tiny, "losses": gen_losses_tiny, "guided": gen_guided_tiny, "helpers": gen_helpers, "clip": gen_clip_tiny, "meantypes": gen_meantypes_tiny,
            "collate": gen_collate, "chunks": gen_chunks_tiny, "real": gen_real_shapes, "negative": gen_negative, "keys": gen_state_dict_keys}
    for name in (args.only.split(",") if args.only else gens):
        gens[name](mods, args.out)
    for f in sorted(os.listdir(args.out)):
        print(f, os.path.getsize(os.path.join(args.out, f)))


if __name__ == "__main__":
    main()
