"""MDM ("V2", w/ cross-local attention + rotary positions): drop-in for reference `model/mdm.py`.

Same constructor keywords (`model/mdm.py:11-13`, as produced by `utils/model_util.py:28-34`),
same `forward(x, timesteps, y)` protocol (`:105-224`), same state-dict keys and shapes
 so reference checkpoints load with `load_model_wo_clip`.  The modules below
only HOLD parameters; all arithmetic runs in libgdx.so (hand-written HIP for gfx950) through
`gesturediffusion_amd.engine.Engine`.  There is no CPU or eager-PyTorch fallback: a CPU tensor
or a missing library raises.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ..engine import GDX_ARCH_MDM, GDX_COND, GDX_UNCOND, Engine, GdxError
from .rotation2xyz import Rotation2xyz

ROPE_ROWS = 4096


def _window_error(n, window):
    msg = f"sequence length must be divisible by window size for local attention: {n} % {window} != 0"
    try:  # the reference fails inside einops.rearrange (model/local_attention.py:104,110)
        from einops import EinopsError
        return EinopsError(msg)
    except Exception:  # noqa: BLE001
        return ValueError(msg)


class PositionalEncoding(nn.Module):
    """Holder of the sinusoidal table (reference model/mdm.py:277-294)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1))   # [max_len, 1, d]


class TimestepEmbedder(nn.Module):
    """reference model/mdm.py:296-310 (shares the PositionalEncoding module, hence the duplicated
    `embed_timestep.sequence_pos_encoder.pe` state-dict key)."""

    def __init__(self, latent_dim, sequence_pos_encoder):
        super().__init__()
        self.latent_dim = latent_dim
        self.sequence_pos_encoder = sequence_pos_encoder
        self.time_embed = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.SiLU(), nn.Linear(latent_dim, latent_dim))


class InputProcess(nn.Module):
    def __init__(self, data_rep, input_feats, latent_dim):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.poseEmbedding = nn.Linear(input_feats, latent_dim)
        if data_rep == "rot_vel":
            self.velEmbedding = nn.Linear(input_feats, latent_dim)


class OutputProcess(nn.Module):
    def __init__(self, data_rep, input_feats, latent_dim, njoints, nfeats):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.njoints, self.nfeats = njoints, nfeats
        self.poseFinal = nn.Linear(latent_dim, input_feats)
        if data_rep == "rot_vel":
            self.velFinal = nn.Linear(latent_dim, input_feats)


class SeedPoseEncoder(nn.Module):
    def __init__(self, njoints, seed_poses, latent_dim):
        super().__init__()
        self.njoints, self.seed_poses, self.latent_dim = njoints, seed_poses, latent_dim
        self.seed_embed = nn.Linear(njoints * seed_poses, latent_dim)


class WavEncoder(nn.Module):
    """Parameter holder only (reference model/mdm.py:312-338); its use raises like the reference."""

    def __init__(self):
        super().__init__()
        self.feat_extractor = nn.Sequential(
            nn.Conv1d(1, 16, 15, stride=5, padding=1600), nn.BatchNorm1d(16), nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(16, 32, 15, stride=5, dilation=4), nn.BatchNorm1d(32), nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(32, 64, 15, stride=5, dilation=7), nn.BatchNorm1d(64), nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(64, 32, 15, stride=5, dilation=13))


class SinusoidalEmbeddings(nn.Module):
    """reference model/local_attention.py:43-53 (buffer `inv_freq`)."""

    def __init__(self, dim):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim)))

    def tables(self, n):
        inv = self.inv_freq.detach().float().cpu()
        t = torch.arange(n).type_as(inv)
        freqs = torch.einsum("i,j->ij", t, inv)
        return freqs.cos().contiguous(), freqs.sin().contiguous()


class SelfAttentionParams(nn.Module):
    """Parameters of nn.MultiheadAttention with its default initialisation."""

    def __init__(self, d, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class EncoderLayerParams(nn.Module):
    """Parameters of one post-norm nn.TransformerEncoderLayer (reference model/mdm.py:90-96)."""

    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = SelfAttentionParams(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class EncoderParams(nn.Module):
    def __init__(self, d, nhead, ff, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayerParams(d, nhead, ff) for _ in range(num_layers)])
        self.num_layers = num_layers


def _norm_dev(device):
    """torch.device with an explicit index ('cuda' -> the current device)."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None and torch.cuda.is_available():
        d = torch.device("cuda", torch.cuda.current_device())
    return d


class _NativeDenoiser(nn.Module):
    """Shared engine plumbing of MDM and MDM_Old."""

    _arch = None

    def _engine_tensors(self):
        named = {k: v for k, v in self.named_parameters()
                 if not k.startswith("clip_model.") and not k.startswith("wav_encoder.") and "velEmbedding" not in k
                 and "velFinal" not in k and not k.startswith("embed_text.")}
        named["sequence_pos_encoder.pe"] = self.sequence_pos_encoder.pe
        return named

    def _new_engine(self, dtype):
        eng = Engine(self._arch, self.input_feats, self.latent_dim, self.ff_size, self.num_layers, self.num_heads,
                     self.seed_poses, cl_head=getattr(self, "cl_head", 8), window=10, compute_dtype=dtype)
        self.__dict__["_eng"] = eng
        self.__dict__["_eng_key"] = None
        return eng

    def _get_engine(self, device):
        named = self._engine_tensors()
        key = tuple((k, v.data_ptr(), v._version) for k, v in named.items())
        eng = self.__dict__.get("_eng")
        dtype = getattr(self, "compute_dtype", None) or os.environ.get("GDX_COMPUTE_DTYPE", "fp32")
        if self.__dict__.get("_packed_image"):
            # weights came from load_packed(): the module's own parameters are NOT the model (they were never loaded), so
            # nothing may be re-packed from them behind the caller's back
            if eng is None or eng.compute_dtype != dtype:
                raise GdxError(f"the weights came from a packed image built for compute_dtype "
                               f"{eng.compute_dtype if eng is not None else '?'!r}; reload the checkpoint (load_state_dict / "
                               f"load_model_cached) to run in {dtype!r}")
            if _norm_dev(device) != self.__dict__.get("_packed_device"):
                raise GdxError(f"the packed image was uploaded to {self.__dict__.get('_packed_device')} but the input is on "
                               f"{device}; load it again on that device")
            return eng
        if eng is not None and eng.compute_dtype != dtype:
            eng = None
        if eng is None:
            eng = self._new_engine(dtype)
        if self.__dict__["_eng_key"] != key:
            for k, v in named.items():
                if v.device != device:
                    raise GdxError(f"parameter {k} is on {v.device} but the input is on {device}; "
                                   f"move the model with .to(device) first")
            extra = self._extra_engine_tensors(device)
            eng.load_tensors({k: v.detach() for k, v in {**named, **extra}.items()})
            self.__dict__["_eng_key"] = key
        return eng

    def _extra_engine_tensors(self, device):
        return {}

    # ---- packed-weight image (include/gdx.h, gdx_export_packed / gdx_import_packed)
    def export_packed(self, device):
        """bytes: the kernels' operand layout of the CURRENT parameters (packs them first if needed)."""
        return self._get_engine(torch.device(device)).export_packed(torch.device(device))

    def load_packed(self, blob, device):
        """Take the weights from a packed image instead of a state dict.  The module's nn.Parameters are left as they are
        and are ignored from here on (state_dict() does NOT describe the loaded model); load_state_dict() switches back to
        them.  Changing compute_dtype (or the device) afterwards raises GdxError at the next call: the image holds one
        dtype's operands and the untouched parameters must not be packed in its place.  Raises GdxError if the image was
        built for another configuration."""
        dtype = getattr(self, "compute_dtype", None) or os.environ.get("GDX_COMPUTE_DTYPE", "fp32")
        eng = self.__dict__.get("_eng")
        if eng is None or eng.compute_dtype != dtype:
            eng = self._new_engine(dtype)
        eng.import_packed(blob, torch.device(device))
        self.__dict__["_packed_image"] = True
        self.__dict__["_packed_device"] = _norm_dev(device)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__["_packed_image"] = False
        self.__dict__["_eng_key"] = None
        return super().load_state_dict(*args, **kwargs)

    def _check_inputs(self, x, y):
        if y is None:
            raise AttributeError("'NoneType' object has no attribute 'get'")   # reference: y.get on None
        if x.dim() != 4:
            raise ValueError("x must be [batch, njoints, nfeats, frames]")
        if x.device.type != "cuda":
            raise GdxError("gesturediffusion_amd runs on MI355X only: x is on %s and there is no CPU fallback" % x.device)
        if self.training and self.cond_mask_prob > 0.0:
            raise NotImplementedError("only the sampling path (model.eval()) is implemented natively")

    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith("clip_model.")]

    def mask_cond(self, cond, force_mask=False):
        """Kept for API parity (reference model/mdm.py:242-250); eval-mode semantics only."""
        if force_mask:
            return torch.zeros_like(cond)
        if self.training and self.cond_mask_prob > 0.0:
            raise NotImplementedError("training-time condition masking is outside the sampling path")
        return cond

    def train(self, *args, **kwargs):
        super().train(*args, **kwargs)
        return self   # the reference forgets `return self` (model/mdm.py:273-275); returning it is a superset


class MDM(_NativeDenoiser):
    _arch = GDX_ARCH_MDM

    def __init__(self, njoints, nfeats, pose_rep, data_rep, latent_dim=256, text_dim=64, ff_size=1024,
                 num_layers=8, num_heads=4, dropout=0.1, activation="gelu", dataset="amass", clip_dim=512,
                 clip_version=None, **kargs):
        super().__init__()
        self.dataset, self.pose_rep, self.data_rep = dataset, pose_rep, data_rep
        self.njoints, self.nfeats = njoints, nfeats
        self.input_feats = njoints * nfeats
        self.latent_dim, self.dropout = latent_dim, dropout
        if activation != "gelu":
            raise NotImplementedError("only activation='gelu' (the reference's hard-coded choice) is implemented")

        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)

        self.use_text = kargs.get("use_text", False)
        self.cond_mask_prob = kargs.get("cond_mask_prob", 0.0)
        self.text_dim, self.clip_dim = text_dim, clip_dim
        if self.use_text:
            raise NotImplementedError("use_text needs CLIP weights (network download); not available")

        self.seed_poses = kargs.get("seed_poses", 0)
        self.compute_dtype = kargs.get("compute_dtype", None)   # additive: "fp32" (default) | "fp16" | "bf16" (engine.COMPUTE_DTYPES)
        if self.seed_poses > 0:
            self.seed_pose_encoder = SeedPoseEncoder(njoints, self.seed_poses, latent_dim)

        self.mfcc_input = kargs.get("mfcc_input", False)
        self.use_wav_enc = kargs.get("use_wav_enc", False)
        if self.mfcc_input:
            self.mfcc_dim = 26
            self.audio_feat_dim = self.mfcc_dim
        if self.use_wav_enc:
            self.wav_enc_dim = 32
            self.audio_feat_dim = self.wav_enc_dim
            self.wav_encoder = WavEncoder()

        self.input_process = InputProcess(data_rep, self.input_feats, latent_dim)
        self.cl_head = 8
        self.project_to_lat = nn.Linear(latent_dim * 2 + self.audio_feat_dim, latent_dim)
        self.rel_pos = SinusoidalEmbeddings(latent_dim // self.cl_head)

        self.num_heads, self.ff_size, self.activation, self.num_layers = num_heads, ff_size, activation, num_layers
        self.seqTransEncoder = EncoderParams(latent_dim, num_heads, ff_size, num_layers)
        self.output_process = OutputProcess(data_rep, self.input_feats, latent_dim, njoints, nfeats)
        self.rot2xyz = Rotation2xyz(device="cpu", dataset=dataset)

    def _extra_engine_tensors(self, device):
        cos, sin = self.rel_pos.tables(ROPE_ROWS)
        return {"rope.cos": cos.to(device), "rope.sin": sin.to(device)}

    def forward(self, x, timesteps, y=None):
        """x [B, njoints, nfeats, T]; timesteps [B] int; y: dict with 'seed' [B,J,1,P], 'mfcc'
        [B,26,1,T], optional 'uncond'.  Returns [B, njoints, nfeats, T] (contiguous)."""
        self._check_inputs(x, y)
        bs, njoints, nfeats, nframes = x.shape
        force_mask = y.get("uncond", False)
        seed = y["seed"]                                   # KeyError like the reference (model/mdm.py:125)
        if self.seed_poses <= 0:
            raise AttributeError("'MDM' object has no attribute 'seed_pose_encoder'")
        if self.mfcc_input:
            mfcc = y["mfcc"]
        elif self.use_wav_enc:
            raise NotImplementedError                      # model/mdm.py:136-137
        else:
            raise NotImplementedError                      # model/mdm.py:139
        if self.data_rep != "genea_vec":
            raise NotImplementedError                      # model/mdm.py:354-358
        if nfeats != 1:
            raise RuntimeError("nfeats must be 1 (the seed encoder flattens njoints * seed_poses)")
        if nframes % 10 != 0:
            raise _window_error(nframes, 10)
        if nframes + 1 > ROPE_ROWS:
            raise ValueError(f"at most {ROPE_ROWS - 1} frames")
        eng = self._get_engine(x.device)
        eng.prepare(bs, nframes)
        eng.set_condition(seed, mfcc)
        out = eng.forward(x, timesteps, GDX_UNCOND if force_mask else GDX_COND)
        return out.view(bs, njoints, nfeats, nframes)
