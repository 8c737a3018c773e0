"""Deterministic weight / input initialiser shared by tests, bench and the golden generator.

There is no network for checkpoints, so every benchmark and parity run uses random
weights of the reference architecture.  The initialiser mimics PyTorch's defaults for
the modules the reference builds (`model/mdm.py:11-103`, `model/mdm_old.py:11-75`):
Linear weight/bias ~ U(+-1/sqrt(fan_in)); MultiheadAttention in_proj_weight Xavier-uniform,
in_proj_bias/out_proj.bias zero; LayerNorm gamma=1, beta=0.  With ``perturb=True`` the
normally trivial tensors (zero biases, unit gammas) get random values so that parity
tests exercise every term.

Each tensor is drawn from its own CPU generator seeded by (seed, crc32(name)), so the
bytes are independent of construction order and identical in every process.
"""
import math
import zlib

import torch

MFCC_DIM = 26


def param_shapes(cfg):
    """State-dict names -> shapes for arch 'mdm' (V2) or 'mdm_old' (V1)."""
    d, ff, J = cfg["latent_dim"], cfg["ff_size"], cfg["njoints"] * cfg["nfeats"]
    sp = cfg["seed_poses"]
    s = {}
    s["embed_timestep.time_embed.0.weight"] = (d, d)
    s["embed_timestep.time_embed.0.bias"] = (d,)
    s["embed_timestep.time_embed.2.weight"] = (d, d)
    s["embed_timestep.time_embed.2.bias"] = (d,)
    s["seed_pose_encoder.seed_embed.weight"] = (d, cfg["njoints"] * sp)
    s["seed_pose_encoder.seed_embed.bias"] = (d,)
    if cfg.get("arch", "mdm") == "mdm":
        s["input_process.poseEmbedding.weight"] = (d, J)
        s["project_to_lat.weight"] = (d, 2 * d + MFCC_DIM)
        s["project_to_lat.bias"] = (d,)
    else:
        s["input_process.poseEmbedding.weight"] = (d, J + MFCC_DIM)
    s["input_process.poseEmbedding.bias"] = (d,)
    for l in range(cfg["num_layers"]):
        p = f"seqTransEncoder.layers.{l}."
        s[p + "self_attn.in_proj_weight"] = (3 * d, d)
        s[p + "self_attn.in_proj_bias"] = (3 * d,)
        s[p + "self_attn.out_proj.weight"] = (d, d)
        s[p + "self_attn.out_proj.bias"] = (d,)
        s[p + "linear1.weight"] = (ff, d)
        s[p + "linear1.bias"] = (ff,)
        s[p + "linear2.weight"] = (d, ff)
        s[p + "linear2.bias"] = (d,)
        s[p + "norm1.weight"] = (d,)
        s[p + "norm1.bias"] = (d,)
        s[p + "norm2.weight"] = (d,)
        s[p + "norm2.bias"] = (d,)
    s["output_process.poseFinal.weight"] = (J, d)
    s["output_process.poseFinal.bias"] = (J,)
    return s


def _gen(seed, name):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def _uniform(shape, bound, g):
    return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound


def init_state_dict(cfg, seed=0, perturb=False):
    sd = {}
    shapes = param_shapes(cfg)
    for name, shape in shapes.items():
        g = _gen(seed, name)
        if name.endswith("in_proj_weight"):
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = _uniform(shape, bound, g)
        elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
            t = _uniform(shape, 0.05, g) if perturb else torch.zeros(shape)
        elif ".norm" in name and name.endswith("weight"):
            t = 1.0 + (_uniform(shape, 0.2, g) if perturb else torch.zeros(shape))
        elif ".norm" in name and name.endswith("bias"):
            t = _uniform(shape, 0.1, g) if perturb else torch.zeros(shape)
        elif name.endswith("weight"):
            t = _uniform(shape, 1.0 / math.sqrt(shape[1]), g)
        else:  # Linear bias: fan_in of the sibling weight
            wname = name[: -len("bias")] + "weight"
            t = _uniform(shape, 1.0 / math.sqrt(shapes[wname][1]), g)
        sd[name] = t.contiguous()
    return sd


def synthetic_inputs(cfg, batch, frames, seed=10):
    """N(0,1) x_T, seed poses and MFCCs (both are z-scored in the reference's dataset:
    data_loaders/gesture/data/dataset.py:77-78,94)."""
    J, nf = cfg["njoints"], cfg["nfeats"]
    g = _gen(seed, "inputs")
    x = torch.randn(batch, J, nf, frames, generator=g)
    seed_poses = torch.randn(batch, J, nf, cfg["seed_poses"], generator=g)
    mfcc = torch.randn(batch, MFCC_DIM, 1, frames, generator=g)
    return x, seed_poses, mfcc
