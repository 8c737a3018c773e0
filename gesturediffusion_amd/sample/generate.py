"""Gesture sampling CLI: `python -m gesturediffusion_amd.sample.generate`.

Keeps the reference CLI's flags and flow (`sample/generate.py:23-183`): args.json override,
fixseed, model + diffusion factory, checkpoint load, optional classifier-free-guidance wrapper,
then `chunks` autoregressive chunks, each a full sampling loop whose seed poses are the last
`seed_poses` frames of the previous chunk (`:104-107`).  Dataset loading and BVH/MP4 writing
(GENEA data, bvhsdk, ffmpeg) are out of scope; with `--synthetic` the conditioning is N(0,1).  The chunk tail of the
reference (`:132-146`: inv_transform with the dataset statistics, position / rotation split; rot2xyz is the identity for
pose_rep 'xyz') runs on the device (`gdx_postprocess`) whenever the feature count is 6 per joint (GENEA: 83 x 6 = 498),
with synthetic statistics; `results.npy` then holds `motion` [B, n_joints, 3, T*chunks] and `motion_rot` like the
reference's, otherwise the normalised poses [B, J, 1, T*chunks].

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N ...`; the batch is
sharded across ranks and gathered once at the end of every chunk (RCCL over xGMI).
"""
import os

import numpy as np
import torch

from .. import engine as E
from ..model.cfg_sampler import ClassifierFreeSampleModel
from ..utils import dist_util
from ..utils.fixseed import fixseed
from ..utils.init import init_state_dict, MFCC_DIM
from ..utils.model_util import create_model_and_diffusion, load_model_wo_clip
from ..utils.parser_util import generate_args


def main(argv=None):
    args = generate_args(argv)
    fixseed(args.seed)
    rank, world, device = dist_util.init_from_env()
    if device.type != "cuda":
        raise RuntimeError("sample.generate needs an MI355X GPU: the native path has no CPU fallback")
    if not args.synthetic:
        raise NotImplementedError("GENEA dataset loading is outside the hot path; use --synthetic "
                                  "(checkpoints still load through --model_path)")
    num_samples = min(args.num_samples if args.num_samples else 41, args.batch_size)
    if args.dataset not in ("genea2022", "genea2023") and not args.synthetic_njoints:
        args.synthetic_njoints = 263
    args.mfcc_input = True if args.synthetic else args.mfcc_input

    model, diffusion = create_model_and_diffusion(args, None)
    if args.model_path:
        state_dict = torch.load(args.model_path, map_location="cpu", weights_only=True)
        load_model_wo_clip(model, state_dict)
    else:
        cfg = dict(arch=args.arch_version, njoints=model.njoints, nfeats=1, latent_dim=args.latent_dim, ff_size=1024,
                   num_layers=args.layers, num_heads=4, seed_poses=args.seed_poses)
        model.load_state_dict(init_state_dict(cfg, seed=args.seed), strict=False)
    if args.guidance_param != 1:
        model = ClassifierFreeSampleModel(model)
    model.to(device)
    model.eval()

    lo, hi = dist_util.shard_range(num_samples, rank, world)
    nb = hi - lo
    J, T = model.njoints, args.num_frames
    g = torch.Generator().manual_seed(args.seed)
    seed_all = torch.randn(num_samples, J, 1, args.seed_poses, generator=g)
    out_chunks, rot_chunks = [], []
    split6 = J % 6 == 0                                   # GENEA layout: 3 rotation + 3 position features per joint
    if split6:
        stat_rng = np.random.default_rng(args.seed)       # stand-in for the dataset's Mean.npy / Std.npy (fp64)
        mean, std = stat_rng.normal(size=J), stat_rng.uniform(0.5, 2.0, size=J)
    sample_fn = diffusion.p_sample_loop if args.sampler == "p" else diffusion.ddim_sample_loop
    sample_out = None
    extractor = None
    if args.synthetic_audio:
        # the reference's audio path (dataset.py:81-95) at its GENEA settings: 22 050 Hz, 30 fps, one MFCC vector per frame
        from ..data_loaders.mfcc import MfccExtractor
        stat = np.random.default_rng(args.seed + 1)
        extractor = MfccExtractor(device, sr=22050, fps=30, mfcc_mean=stat.normal(size=MFCC_DIM),
                                  mfcc_std=stat.uniform(0.5, 2.0, size=MFCC_DIM))
    for chunk in range(args.chunks):
        if extractor is not None:
            audio = 0.1 * torch.randn(num_samples, T * 735, generator=g)[lo:hi].to(device)
            mfcc = torch.stack([extractor(a)[:T].t() for a in audio]).unsqueeze(2).contiguous()     # [nb, 26, 1, T]
        else:
            mfcc = torch.randn(num_samples, MFCC_DIM, 1, T, generator=g)[lo:hi].to(device)
        y = {"mfcc": mfcc, "seed": seed_all[lo:hi].to(device) if chunk == 0 else sample_out[..., -args.seed_poses:]}
        if args.guidance_param != 1:
            y["scale"] = torch.ones(nb, device=device) * args.guidance_param
        kw = dict(clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None, progress=False,
                  noise=None, rng=args.rng, philox_seed=args.seed + 1000 * chunk, sample_offset=lo)
        if args.sampler == "p":
            kw.update(dump_steps=None, const_noise=False)
        else:
            kw.update(eta=args.eta)
        if rank == 0:
            print(f"### Sampling chunk {chunk + 1} of {args.chunks}")
        sample_out = sample_fn(model, (nb, J, 1, T), **kw)
        full = dist_util.gather_samples(sample_out, num_samples)
        if rank == 0:
            if split6:
                pos, rot = E.postprocess(full, mean, std)   # inv_transform + index split on the device
                out_chunks.append(pos.cpu().numpy())
                rot_chunks.append(rot.cpu().numpy())
            else:
                out_chunks.append(full.cpu().numpy())
    if rank == 0:
        out_path = args.output_dir or os.path.join(os.getcwd(), f"samples_synthetic_seed{args.seed}")
        os.makedirs(out_path, exist_ok=True)
        motion = np.concatenate(out_chunks, axis=3)
        res = {"motion": motion, "num_samples": num_samples, "num_chunks": args.chunks}
        if split6:
            res["motion_rot"] = np.concatenate(rot_chunks, axis=3)
        np.save(os.path.join(out_path, "results.npy"), res, allow_pickle=True)
        print(f"saved results to [{os.path.join(out_path, 'results.npy')}] motion {motion.shape}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
