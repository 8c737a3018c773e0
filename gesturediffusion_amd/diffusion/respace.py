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
    """Set of original timesteps to keep.  "ddimN": the first integer stride that yields exactly N steps; otherwise
    a list (or comma-separated string) of per-section counts, each section sampled at an even fractional stride whose
    running position is accumulated in floating point and rounded (the accumulation order decides ties)."""
    if isinstance(section_counts, str) and section_counts.startswith("ddim"):
        want = int(section_counts[4:])
        stride = next((s for s in range(1, num_timesteps) if -(-num_timesteps // s) == want), None)
        if stride is None:
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        return set(range(0, num_timesteps, stride))
    counts = [int(c) for c in section_counts.split(",")] if isinstance(section_counts, str) else list(section_counts)
    base, extra = divmod(num_timesteps, len(counts))
    kept, first = set(), 0
    for k, n in enumerate(counts):
        size = base + (1 if k < extra else 0)
        if size < n:
            raise ValueError(f"cannot divide section of {size} steps into {n}")
        step = (size - 1) / (n - 1) if n > 1 else 1
        pos = 0.0
        for _ in range(n):
            kept.add(first + round(pos))
            pos += step
        first += size
    return kept


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process over the kept subset of a base process's steps: beta_k = 1 - abar[t_k] / abar[t_{k-1}]."""

    def __init__(self, use_timesteps, **kwargs):
        full = GaussianDiffusion(**kwargs)
        self.original_num_steps = full.num_timesteps
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = sorted(t for t in self.use_timesteps if 0 <= t < full.num_timesteps)
        abar = full.alphas_cumprod[self.timestep_map]
        kwargs["betas"] = 1 - abar / np.concatenate(([1.0], abar[:-1]))
        super().__init__(**kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps,
                             self.__dict__.setdefault("_map_tables", {}))

    def _call_model(self, model, x, t, model_kwargs):
        return self._wrap_model(model)(x, t, **model_kwargs)

    def _call_cond_fn(self, cond_fn, x, t, model_kwargs):
        return self._wrap_model(cond_fn)(x, t, **model_kwargs)        # reference respace.py:99-103

    def _timestep_map(self):
        return list(self.timestep_map)

    def _scale_timesteps(self, t):
        return t   # scaling is done by the wrapped model


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps, tables=None):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        # device copies of the map per (device, dtype); owned by the SpacedDiffusion that built this wrapper (so a
        # new wrapper per step reuses them) and never shared between diffusions -- two maps of equal length
        # ("ddim10": 0,100,..,900 vs [10]: 0,111,..,999) must not see each other's table
        self._tables = tables if tables is not None else {}

    def __call__(self, x, ts, **kwargs):
        return self.model(x, self.map_timesteps(ts), **kwargs)

    def map_timesteps(self, ts):
        """Respaced index -> original timestep, looked up on `ts`'s device."""
        key = (str(ts.device), ts.dtype)
        table = self._tables.get(key)
        if table is None:
            table = self._tables[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        mapped = table[ts]
        if self.rescale_timesteps:
            mapped = mapped.float() * (1000.0 / self.original_num_steps)
        return mapped
