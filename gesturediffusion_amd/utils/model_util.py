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


def load_model_wo_clip(model, state_dict):
    missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
    assert len(unexpected_keys) == 0
    assert all([k.startswith('clip_model.') for k in missing_keys])


def create_model_and_diffusion(args, data=None):
    cls = MDM_Old if getattr(args, "arch_version", "mdm") == "mdm_old" else MDM
    model = cls(**get_model_args(args, data), compute_dtype=getattr(args, "compute_dtype", None))
    diffusion = create_gaussian_diffusion(args)
    return model, diffusion


def get_model_args(args, data=None):
    clip_version = 'ViT-B/32'
    if args.dataset in ['genea2022', 'genea2023']:
        data_rep, njoints, nfeats = 'genea_vec', 498, 1
    elif getattr(args, "synthetic_njoints", None):
        # additive: synthetic shapes for benchmarking (HumanML3D 263, HumanAct12 25x6=150 flattened)
        data_rep, njoints, nfeats = 'genea_vec', int(args.synthetic_njoints), 1
    else:
        raise UnboundLocalError("local variable 'data_rep' referenced before assignment")  # as the reference
    return {'modeltype': '', 'njoints': njoints, 'nfeats': nfeats, 'translation': True, 'pose_rep': 'rot6d',
            'glob': True, 'glob_rot': True, 'latent_dim': args.latent_dim, 'ff_size': 1024,
            'num_layers': args.layers, 'num_heads': 4, 'dropout': 0.1, 'activation': "gelu", 'data_rep': data_rep,
            'cond_mask_prob': args.cond_mask_prob, 'clip_version': clip_version, 'dataset': args.dataset,
            'use_text': args.use_text, 'mfcc_input': args.mfcc_input, 'use_wav_enc': args.use_wav_enc,
            'seed_poses': args.seed_poses, 'use_audio': args.use_audio}


def create_gaussian_diffusion(args):
    predict_xstart = True
    steps = 1000
    scale_beta = 1.0
    timestep_respacing = getattr(args, "timestep_respacing", '') or ''
    learn_sigma = False
    rescale_timesteps = False
    betas = gd.get_named_beta_schedule(args.noise_schedule, steps, scale_beta)
    loss_type = gd.LossType.MSE
    if not timestep_respacing:
        timestep_respacing = [steps]
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing),
        betas=betas,
        model_mean_type=(gd.ModelMeanType.EPSILON if not predict_xstart else gd.ModelMeanType.START_X),
        model_var_type=((gd.ModelVarType.FIXED_LARGE if not args.sigma_small else gd.ModelVarType.FIXED_SMALL)
                        if not learn_sigma else gd.ModelVarType.LEARNED_RANGE),
        loss_type=loss_type,
        rescale_timesteps=rescale_timesteps,
        lambda_vel=args.lambda_vel,
        lambda_rcxyz=args.lambda_rcxyz,
        lambda_fc=args.lambda_fc,
    )
