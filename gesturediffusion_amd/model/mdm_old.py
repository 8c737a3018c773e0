"""MDM_Old ("V1", encoder-only topology): drop-in for reference `model/mdm_old.py`.

Same constructor keywords (`model/mdm_old.py:12-14`), `forward(x, timesteps, y)` protocol
(`:84-122`) and state-dict layout.  This is the topology BASELINE.json's headline config
(T = 196) runs, because V2's local attention requires T % 10 == 0.
All arithmetic runs in libgdx.so; the modules only hold parameters.
"""
import torch.nn as nn

from ..engine import GDX_ARCH_MDM_OLD, GDX_COND, GDX_UNCOND
from .mdm import (EncoderParams, InputProcess, OutputProcess, PositionalEncoding, SeedPoseEncoder, TimestepEmbedder,
                  _NativeDenoiser)
from .rotation2xyz import Rotation2xyz


class MDM_Old(_NativeDenoiser):
    _arch = GDX_ARCH_MDM_OLD

    def __init__(self, njoints, nfeats, translation, pose_rep, glob, glob_rot, latent_dim=256, ff_size=1024,
                 num_layers=8, num_heads=4, dropout=0.1, activation="gelu", data_rep="rot6d", dataset="amass",
                 **kargs):
        super().__init__()
        self.dataset, self.pose_rep = dataset, pose_rep
        self.njoints, self.nfeats = njoints, nfeats
        self.input_feats = njoints * nfeats
        self.latent_dim = latent_dim
        self.cond_mode = kargs.get("cond_mode", "no_cond")
        self.glob, self.glob_rot, self.translation = glob, glob_rot, translation
        if activation != "gelu":
            raise NotImplementedError("only activation='gelu' is implemented")

        self.cond_mask_prob = kargs.get("cond_mask_prob", 0.0)
        self.seed_poses = kargs.get("seed_poses", 0)
        self.compute_dtype = kargs.get("compute_dtype", None)   # additive: "fp32" (default) | "fp16"
        assert self.seed_poses > 0                                   # model/mdm_old.py:32
        self.seed_pose_encoder = SeedPoseEncoder(njoints, self.seed_poses, latent_dim)
        self.mfcc_dim = 26
        self.data_rep = data_rep
        self.input_process = InputProcess(data_rep, self.input_feats + self.mfcc_dim, latent_dim)

        self.num_heads, self.ff_size, self.dropout = num_heads, ff_size, dropout
        self.activation, self.num_layers = activation, num_layers
        self.seqTransEncoder = EncoderParams(latent_dim, num_heads, ff_size, num_layers)
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)
        self.output_process = OutputProcess(data_rep, self.input_feats, latent_dim, njoints, nfeats)
        self.rot2xyz = Rotation2xyz(device="cpu", dataset=dataset)

    def forward(self, x, timesteps, y=None):
        self._check_inputs(x, y)
        bs, njoints, nfeats, nframes = x.shape
        force_mask = y.get("uncond", False)
        seed = y["seed"]
        mfcc = y["mfcc"]
        if self.data_rep != "genea_vec":
            raise NotImplementedError                                # model/mdm_old.py InputProcess
        if nfeats != 1:
            raise RuntimeError("nfeats must be 1 (the seed encoder flattens njoints * seed_poses)")
        if nframes + 1 > self.sequence_pos_encoder.pe.shape[0]:
            raise ValueError("sequence longer than the positional table")
        eng = self._get_engine(x.device)
        eng.prepare(bs, nframes)
        eng.set_condition(seed, mfcc)
        out = eng.forward(x, timesteps, GDX_UNCOND if force_mask else GDX_COND)
        return out.view(bs, njoints, nfeats, nframes)
