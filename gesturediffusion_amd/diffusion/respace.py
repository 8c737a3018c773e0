"""Timestep respacing: drop-in for reference `diffusion/respace.py` (space_timesteps :8-61,
SpacedDiffusion :64-114, _WrappedModel :117-129).

Difference in mechanism, not behaviour: the reference rebuilds `th.tensor(timestep_map, device)`
(a host->device copy and sync) on every model call; here the map lives on the device once per
(device) and the fused loop hands it to libgdx, which looks timesteps up on the GPU.
"""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired_count = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired_count:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx = 0
    all_steps = []
    for i, section_count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < section_count:
            raise ValueError(f"cannot divide section of {size} steps into {section_count}")
        frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
        cur_idx = 0.0
        taken_steps = []
        for _ in range(section_count):
            taken_steps.append(start_idx + round(cur_idx))
            cur_idx += frac_stride
        all_steps += taken_steps
        start_idx += size
    return set(all_steps)


class SpacedDiffusion(GaussianDiffusion):
    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base_diffusion = GaussianDiffusion(**kwargs)
        last_alpha_cumprod = 1.0
        new_betas = []
        for i, alpha_cumprod in enumerate(base_diffusion.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - alpha_cumprod / last_alpha_cumprod)
                last_alpha_cumprod = alpha_cumprod
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _call_model(self, model, x, t, model_kwargs):
        return self._wrap_model(model)(x, t, **model_kwargs)

    def _call_cond_fn(self, cond_fn, x, t, model_kwargs):
        return self._wrap_model(cond_fn)(x, t, **model_kwargs)        # reference respace.py:99-103

    def _timestep_map(self):
        return list(self.timestep_map)

    def _scale_timesteps(self, t):
        return t   # scaling is done by the wrapped model


_MAP_CACHE = {}


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps

    def __call__(self, x, ts, **kwargs):
        key = (id(self.timestep_map), len(self.timestep_map), str(ts.device), ts.dtype)
        map_tensor = _MAP_CACHE.get(key)
        if map_tensor is None:
            map_tensor = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
            if len(_MAP_CACHE) > 64:
                _MAP_CACHE.clear()
            _MAP_CACHE[key] = map_tensor
        new_ts = map_tensor[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
