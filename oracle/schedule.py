"""Oracle: noise schedule tables in numpy fp64 (test infrastructure only).

Restates reference `diffusion/gaussian_diffusion.py:20-64` (named beta schedules),
`:164-199` (derived tables) and `diffusion/respace.py:8-61,73-87` (timestep
respacing).  Pinned by tests/golden/schedule.npz and the known answers listed in
SURVEY.md section 8a (A1, A2).
"""
import math

import numpy as np


def betas_for_alpha_bar(n, alpha_bar, max_beta=0.999):
    # reference diffusion/gaussian_diffusion.py:47-64
    out = np.empty(n, dtype=np.float64)
    for i in range(n):
        t1 = i / n
        t2 = (i + 1) / n
        out[i] = min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta)
    return out


def named_beta_schedule(name, n, scale_betas=1.0):
    # reference diffusion/gaussian_diffusion.py:20-44
    if name == "linear":
        scale = scale_betas * 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "cosine":
        return betas_for_alpha_bar(n, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def space_timesteps(num_timesteps, section_counts):
    # reference diffusion/respace.py:8-61
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start = 0
    steps = []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


class Tables:
    """fp64 tables of reference GaussianDiffusion.__init__ (gaussian_diffusion.py:164-199)."""

    def __init__(self, betas):
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)


def respace(base_betas, use_timesteps):
    """reference diffusion/respace.py:73-87: betas of the kept steps + timestep_map."""
    base = Tables(base_betas)
    use = set(use_timesteps)
    last = 1.0
    new_betas, tmap = [], []
    for i, ac in enumerate(base.alphas_cumprod):
        if i in use:
            new_betas.append(1 - ac / last)
            last = ac
            tmap.append(i)
    return np.array(new_betas), tmap


def make_tables(noise_schedule="cosine", steps=1000, respacing=""):
    """reference utils/model_util.py:37-72 (create_gaussian_diffusion) minus loss plumbing."""
    betas = named_beta_schedule(noise_schedule, steps)
    if not respacing:
        respacing = [steps]
    new_betas, tmap = respace(betas, space_timesteps(steps, respacing))
    return Tables(new_betas), tmap
