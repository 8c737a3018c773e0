"""Classifier-free-guidance sampling wrapper: drop-in for reference `model/cfg_sampler.py`.

The reference runs the inner model twice per step (cond, then a deep copy of `y` with
`uncond=True`) and blends `u + scale * (c - u)` (`model/cfg_sampler.py:23-28`).  Here both
passes run as ONE double batch through the HIP kernels (only the seed-pose embedding differs
between them; the MFCC conditioning feeds both, as in the reference) and the blend is a fused
kernel.  Nothing is deep-copied.
"""
import torch.nn as nn

from ..engine import GDX_CFG
from .mdm import _NativeDenoiser


class ClassifierFreeSampleModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        assert self.model.cond_mask_prob > 0, \
            'Cannot run a guided diffusion on a model that has not been trained with no conditions'
        if not isinstance(model, _NativeDenoiser):
            raise TypeError("ClassifierFreeSampleModel wraps gesturediffusion_amd MDM / MDM_Old instances")
        self.rot2xyz = self.model.rot2xyz
        self.njoints = self.model.njoints
        self.nfeats = self.model.nfeats
        self.data_rep = self.model.data_rep

    def forward(self, x, timesteps, y=None):
        m = self.model
        m._check_inputs(x, y)
        bs, njoints, nfeats, nframes = x.shape
        scale = y["scale"]
        y_c = dict(y)
        y_c.pop("uncond", None)
        # argument validation identical to a plain forward (raises the same errors)
        if getattr(m, "_arch", None) is None:
            raise TypeError("inner model has no native engine")
        seed, mfcc = y["seed"], y["mfcc"]
        if hasattr(m, "cl_head") and nframes % 10 != 0:
            from .mdm import _window_error
            raise _window_error(nframes, 10)
        if m.data_rep != "genea_vec":
            raise NotImplementedError
        eng = m._get_engine(x.device)
        eng.prepare(bs, nframes)
        eng.set_condition(seed, mfcc)
        out = eng.forward(x, timesteps, GDX_CFG, scale.reshape(-1))
        return out.view(bs, njoints, nfeats, nframes)
