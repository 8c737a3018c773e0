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
sharded across ranks and gathered once at the end of every chunk (RCCL over xGMI).  Sharded runs draw their noise from
the counter-based generator keyed by the global sample index (`resolve_rng`), so the result does not depend on N.
"""
import os

import numpy as np
import torch

from .. import engine as E
from ..model.cfg_sampler import ClassifierFreeSampleModel
from ..utils import dist_util
from ..utils.fixseed import fixseed
from ..utils.init import init_state_dict, MFCC_DIM
from ..utils.model_util import create_model_and_diffusion, load_checkpoint, load_model_cached, load_model_wo_clip
from ..utils.parser_util import generate_args


def resolve_rng(rng, world):
    """Noise source of a run.  torch's generator (the reference's) gives every rank the same stream, so sharded samples
    would be correlated and depend on the world size: multi-GPU runs use the counter-based generator keyed by the
    GLOBAL sample index (shard invariant); asking for `--rng torch` there is an error, not a silent change."""
    if world > 1:
        if rng == "torch":
            raise ValueError("--rng torch with WORLD_SIZE > 1: every rank would draw the same noise for its shard; "
                             "use --rng philox (the multi-GPU default)")
        return "philox"
    return rng or "torch"


def sample_chunks(model, diffusion, first_seed, mfcc_of_chunk, n_chunks, frames, seed_poses, guidance_param=1.0,
                  sampler="p", eta=0.0, rng="torch", philox_seed=0, sample_offset=0, noise_tapes=None, progress=False,
                  on_chunk=None):
    """The chunked autoregressive driver of reference `sample/generate.py:91-130`: chunk c is one complete sampling loop
    conditioned on its MFCCs and on seed poses that are `first_seed` for c = 0 and afterwards the LAST `seed_poses` frames
    of chunk c-1 -- a view of the previous output that stays on the device (`:104-107`).  Yields nothing; returns the list
    of chunk outputs [b, J, 1, frames].  noise_tapes: optional list of recorded noise tapes, one per chunk (tests)."""
    b, J = first_seed.shape[0], first_seed.shape[1]
    sample_fn = diffusion.p_sample_loop if sampler == "p" else diffusion.ddim_sample_loop
    outs, sample_out = [], None
    for chunk in range(n_chunks):
        y = {"mfcc": mfcc_of_chunk(chunk), "seed": first_seed if chunk == 0 else sample_out[..., -seed_poses:]}
        if guidance_param != 1:
            y["scale"] = torch.ones(b, device=first_seed.device) * guidance_param
        kw = dict(clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None, progress=progress,
                  noise=None, rng=rng, philox_seed=philox_seed + 1000 * chunk, sample_offset=sample_offset,
                  noise_tape=noise_tapes[chunk] if noise_tapes is not None else None)
        if sampler == "p":
            kw.update(dump_steps=None, const_noise=False)
        else:
            kw.update(eta=eta)
        if on_chunk is not None:
            on_chunk(chunk)
        sample_out = sample_fn(model, (b, J, 1, frames), **kw)
        outs.append(sample_out)
    return outs


def main(argv=None):
    args = generate_args(argv)
    fixseed(args.seed)
    rank, world, device = dist_util.init_from_env()
    if device.type != "cuda":
        raise RuntimeError("sample.generate needs an MI355X GPU: the native path has no CPU fallback")
    if not args.synthetic:
        raise NotImplementedError("GENEA dataset loading is outside the hot path; use --synthetic "
                                  "(checkpoints still load through --model_path)")
    rng = resolve_rng(args.rng, world)
    num_samples = min(args.num_samples if args.num_samples else 41, args.batch_size)
    dist_util.check_world(num_samples, world)       # the same refusal on every rank, before any of them builds a model
    if args.dataset not in ("genea2022", "genea2023") and not args.synthetic_njoints:
        args.synthetic_njoints = 263
    args.mfcc_input = True if args.synthetic else args.mfcc_input

    model, diffusion = create_model_and_diffusion(args, None)
    if args.model_path and args.packed_cache:
        model.to(device)
        how = load_model_cached(model, args.model_path, device, args.packed_cache)
        if rank == 0:
            print(f"### weights from the packed {how}" if how == "image" else "### weights from the checkpoint (packed image written)")
    elif args.model_path:
        state_dict = load_checkpoint(args.model_path)
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
    J, T = model.njoints, args.num_frames
    g = torch.Generator().manual_seed(args.seed)
    seed_all = torch.randn(num_samples, J, 1, args.seed_poses, generator=g)
    split6 = J % 6 == 0                                   # GENEA layout: 3 rotation + 3 position features per joint
    if split6:
        stat_rng = np.random.default_rng(args.seed)       # stand-in for the dataset's Mean.npy / Std.npy (fp64)
        mean, std = stat_rng.normal(size=J), stat_rng.uniform(0.5, 2.0, size=J)
    extractor = None
    if args.synthetic_audio:
        # the reference's audio path (dataset.py:81-95) at its GENEA settings: 22 050 Hz, 30 fps, one MFCC vector per frame
        from ..data_loaders.mfcc import MfccExtractor
        stat = np.random.default_rng(args.seed + 1)
        extractor = MfccExtractor(device, sr=22050, fps=30, mfcc_mean=stat.normal(size=MFCC_DIM),
                                  mfcc_std=stat.uniform(0.5, 2.0, size=MFCC_DIM))

    def mfcc_of_chunk(chunk):          # called once per chunk, in order: the host generator's stream is part of the recipe
        if extractor is not None:
            audio = 0.1 * torch.randn(num_samples, T * 735, generator=g)[lo:hi].to(device)
            return torch.stack([extractor(a)[:T].t() for a in audio]).unsqueeze(2).contiguous()     # [nb, 26, 1, T]
        return torch.randn(num_samples, MFCC_DIM, 1, T, generator=g)[lo:hi].to(device)

    def on_chunk(chunk):
        if rank == 0:
            print(f"### Sampling chunk {chunk + 1} of {args.chunks}")

    outs = sample_chunks(model, diffusion, seed_all[lo:hi].to(device), mfcc_of_chunk, args.chunks, T, args.seed_poses,
                         guidance_param=args.guidance_param, sampler=args.sampler, eta=args.eta, rng=rng,
                         philox_seed=args.seed, sample_offset=lo, progress=args.progress and rank == 0, on_chunk=on_chunk)
    out_chunks, rot_chunks = [], []
    for sample_out in outs:
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
