"""Model / diffusion factory: drop-in for reference `utils/model_util.py`.

`create_model_and_diffusion(args, data)` builds the native MDM and a SpacedDiffusion with the
reference's fixed choices (START_X, cosine/linear schedule, 1000 steps, FIXED_SMALL unless
`sigma_small` is false; `utils/model_util.py:37-72`).  Additive: `args.timestep_respacing`
(e.g. "ddim100"), which the reference hard-codes to '' (`:42`), and `args.arch_version`
("mdm" = V2 default, "mdm_old" = V1).
"""
from ..diffusion import gaussian_diffusion as gd
from ..diffusion.respace import SpacedDiffusion, space_timesteps
from ..model.mdm import MDM
from ..model.mdm_old import MDM_Old


DIFFUSION_STEPS = 1000          # fixed: the reference parses --diffusion_steps and then ignores it (utils/model_util.py:40)


def load_checkpoint(path):
    """A `model#########.pt` written by the reference's trainer (`train/training_loop.py:265-285`: a plain
    `torch.save(state_dict)` without the CLIP weights).  Loaded with `weights_only=True`: nothing in the file executes."""
    import torch
    return torch.load(path, map_location="cpu", weights_only=True)


def packed_image_path(cache_dir, checkpoint_path, model):
    """File name of the packed-weight image of `checkpoint_path` for `model`'s architecture and compute dtype: keyed by the
    checkpoint's content (sha256), so a re-trained file of the same name never matches a stale image."""
    import hashlib
    import os
    h = hashlib.sha256()
    with open(checkpoint_path, "rb") as f:
        for block in iter(lambda: f.read(1 << 22), b""):
            h.update(block)
    dtype = getattr(model, "compute_dtype", None) or os.environ.get("GDX_COMPUTE_DTYPE", "fp32")
    return os.path.join(cache_dir, f"{h.hexdigest()[:24]}-{type(model).__name__}-{dtype}.gdxpack")


def load_model_cached(model, checkpoint_path, device, cache_dir):
    """Checkpoint ingestion through the weight pre-packing cache.  First use of a checkpoint: the normal
    path (`load_checkpoint` + `load_model_wo_clip`), then the packed operand layout is written to `cache_dir`.  Later uses:
    the image is uploaded as it is -- no unpickling, no per-tensor repack; the module's nn.Parameters stay untouched
    (`MDM.load_packed`).  Returns "image" or "checkpoint" (which path ran).  `model` must already be on `device`."""
    import os
    path = packed_image_path(cache_dir, checkpoint_path, model)
    if os.path.exists(path):
        with open(path, "rb") as f:
            blob = f.read()
        try:
            model.load_packed(blob, device)
            return "image"
        except Exception as e:  # noqa: BLE001 - a stale / foreign / truncated image is not fatal: rebuild it
            print(f"[packed cache] ignoring {path}: {e}")
    load_model_wo_clip(model, load_checkpoint(checkpoint_path))
    blob = model.export_packed(device)
    os.makedirs(cache_dir, exist_ok=True)
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(blob)
    os.replace(tmp, path)                                  # atomic: concurrent ranks never read a half-written image
    return "checkpoint"


def load_model_wo_clip(model, state_dict):
    """Non-strict load with the reference's two guarantees (`utils/model_util.py:6-9`, both AssertionError): the checkpoint
    holds no key the model does not know, and only CLIP weights may be absent from it."""
    outcome = model.load_state_dict(state_dict, strict=False)
    assert not outcome.unexpected_keys, f"unexpected keys in checkpoint: {outcome.unexpected_keys[:5]}"
    stray = [k for k in outcome.missing_keys if not k.startswith("clip_model.")]
    assert not stray, f"checkpoint lacks non-CLIP parameters: {stray[:5]}"


def create_model_and_diffusion(args, data=None):
    cls = MDM_Old if getattr(args, "arch_version", "mdm") == "mdm_old" else MDM
    model = cls(**get_model_args(args, data), compute_dtype=getattr(args, "compute_dtype", None))
    return model, create_gaussian_diffusion(args)


def get_model_args(args, data=None):
    """Constructor keywords of MDM / MDM_Old.  The key set is the contract with `MDM.__init__(**kargs)`
    (`utils/model_util.py:18-34`); everything not taken from `args` is a constant there too."""
    if args.dataset in ("genea2022", "genea2023"):
        njoints = 498                                    # 83 joints x (3 rotation + 3 position) features
    elif getattr(args, "synthetic_njoints", None):
        njoints = int(args.synthetic_njoints)            # additive: synthetic shapes (HumanML3D 263, HumanAct12 150)
    else:
        # the reference falls off its if-chain with data_rep unbound; same exception class for callers that catch it
        raise UnboundLocalError("local variable 'data_rep' referenced before assignment")
    fixed = dict(modeltype="", nfeats=1, translation=True, pose_rep="rot6d", glob=True, glob_rot=True, ff_size=1024,
                 num_heads=4, dropout=0.1, activation="gelu", data_rep="genea_vec", clip_version="ViT-B/32")
    from_args = dict(njoints=njoints, latent_dim=args.latent_dim, num_layers=args.layers,
                     cond_mask_prob=args.cond_mask_prob, dataset=args.dataset, use_text=args.use_text,
                     mfcc_input=args.mfcc_input, use_wav_enc=args.use_wav_enc, seed_poses=args.seed_poses,
                     use_audio=args.use_audio)
    return {**fixed, **from_args}


def create_gaussian_diffusion(args):
    """The single sampler configuration the reference builds (`utils/model_util.py:37-72`): the network predicts x_0,
    the reverse variance is fixed (the posterior variance unless `sigma_small` is falsy, then beta), the loss is plain
    MSE, timesteps are passed to the model unscaled, and the schedule has 1000 steps.  `args.timestep_respacing` is
    additive (the reference hard-codes ''): '' keeps every step."""
    respacing = getattr(args, "timestep_respacing", "") or [DIFFUSION_STEPS]
    variance = gd.ModelVarType.FIXED_SMALL if args.sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(use_timesteps=space_timesteps(DIFFUSION_STEPS, respacing),
                           betas=gd.get_named_beta_schedule(args.noise_schedule, DIFFUSION_STEPS),
                           model_mean_type=gd.ModelMeanType.START_X, model_var_type=variance,
                           loss_type=gd.LossType.MSE, rescale_timesteps=False, lambda_vel=args.lambda_vel,
                           lambda_rcxyz=args.lambda_rcxyz, lambda_fc=args.lambda_fc)
