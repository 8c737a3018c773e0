"""Gaussian diffusion *sampling* process: drop-in for the hot subset of reference
`diffusion/gaussian_diffusion.py` (schedule :20-64, tables :120-199, q_sample :233-251,
p_mean_variance :277-388, p_sample :496-548, p_sample_loop{,_progressive} :598-730,
ddim_sample :732-782, ddim_sample_loop{,_progressive} :879-993).

Host side only builds the fp64 schedule tables (numpy, like the reference) and the per-step fp32
coefficient rows; every per-element operation runs in libgdx.so:

  * `gdx_forward`         the denoiser (through the model callable protocol `model(x, ts, **kw)`)
  * `gdx_sampler_update`  CFG blend + inpainting + posterior mean / DDIM step + noise, one pass
  * `gdx_sample_loop`     the whole loop enqueued from C++ with in-kernel Philox noise

Configured mode = the one the reference hard-codes (`utils/model_util.py:37-72`): START_X mean, FIXED_SMALL / FIXED_LARGE
variance; this is what `gdx_sample_loop` runs.  The other two readings of the denoiser output, EPSILON and PREVIOUS_X
(`:357-372`), go through the step-wise protocol (one extra element-wise launch per step).  The backward half of training
and cond_fn_with_grad (autograd through the denoiser) raise NotImplementedError, and so do learned variances: they need a
denoiser with twice the channels, which neither MDM has (the reference's own assert at `:317` fires for them).

RNG.  `rng="torch"` (default) draws x_T with `torch.randn` and one N(0,1) tensor per step from torch's
generator on the sample's device, in the reference's order (`:694`, `:532`), so a run is reproducible against
the reference on the same device and seed; the draws are made `NOISE_BLOCK` steps ahead into a tape and the
loop itself runs inside libgdx (`gdx_sample_loop`, one call per block), so the reference caller's own kwargs
(`sample/generate.py:119-130`: `progress=True, noise=None`) take the fused path.  `rng="philox"` generates the
noise inside the update kernel, counter-based and keyed by (seed, global sample index, step): results do not
depend on how a batch is sharded over GPUs.  `noise_tape=` replays recorded noise (parity tests).
"""
import enum
import math

import numpy as np
import torch as th

from .. import engine as E
from ..engine import GDX_CFG, GDX_COND, GDX_UNCOND
from .._lib import GDX_SAMPLER_DDIM, GDX_SAMPLER_P


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.0):
    if schedule_name == "linear":
        scale = scale_betas * 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    betas = []
    for i in range(num_diffusion_timesteps):
        t1 = i / num_diffusion_timesteps
        t2 = (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self == LossType.KL or self == LossType.RESCALED_KL


NOISE_BLOCK = 50        # most steps of torch-generator noise drawn ahead of one gdx_sample_loop call (rng="torch") ...
NOISE_BLOCK_BYTES = 256 << 20   # ... within this many bytes of tape (660 MB at config 2 and 6.6 GB at config 5 otherwise)


def noise_block_steps(n_steps, per_step_elems):
    """Steps of pre-drawn noise per gdx_sample_loop call: at most NOISE_BLOCK, at most NOISE_BLOCK_BYTES of fp32 tape, at least one.
    Blocks are issued back to back without a host synchronisation, so smaller blocks cost nothing measurable."""
    by_bytes = NOISE_BLOCK_BYTES // max(1, 4 * int(per_step_elems))
    return max(1, min(int(n_steps), NOISE_BLOCK, by_bytes))


def _is_native(model):
    from ..model.cfg_sampler import ClassifierFreeSampleModel
    from ..model.mdm import _NativeDenoiser
    return isinstance(model, (_NativeDenoiser, ClassifierFreeSampleModel))


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False,
                 lambda_rcxyz=0.0, lambda_vel=0.0, lambda_pose=1.0, lambda_orient=1.0, lambda_loc=1.0,
                 data_rep="rot6d", lambda_root_vel=0.0, lambda_vel_rcxyz=0.0, lambda_fc=0.0):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        self.data_rep = data_rep
        if data_rep != "rot_vel" and lambda_pose != 1.0:
            raise ValueError("lambda_pose is relevant only when training on velocities!")
        self.lambda_pose, self.lambda_orient, self.lambda_loc = lambda_pose, lambda_orient, lambda_loc
        self.lambda_rcxyz, self.lambda_vel, self.lambda_root_vel = lambda_rcxyz, lambda_vel, lambda_root_vel
        self.lambda_vel_rcxyz, self.lambda_fc = lambda_vel_rcxyz, lambda_fc

        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])

        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        assert self.alphas_cumprod_prev.shape == (self.num_timesteps,)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self._coef_cache = {}

    # ------------------------------------------------------------------ coefficient rows
    def _model_variance_tables(self):
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            v = np.append(self.posterior_variance[1], self.betas[1:])
            return v, np.log(v)
        if self.model_var_type == ModelVarType.FIXED_SMALL:
            return self.posterior_variance, self.posterior_log_variance_clipped
        raise NotImplementedError("learned variances are outside the sampling hot path")

    def _check_supported(self):
        """START_X (the reference's configuration; the only reading the fused loop takes), EPSILON and PREVIOUS_X
        (step-wise protocol) with fixed variances.  Learned variances need a denoiser with 2x the channels, which neither
        MDM nor MDM_Old has (the reference's own assert at :317 fires for them)."""
        self._model_variance_tables()

    def _xstart_table(self, device):
        """Rows (c0, c1) of gdx_plms_update kind 8 for the configured reading of the denoiser output, rounded like every
        other table (fp64 -> .float()): EPSILON (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod), reference
        :390-396; PREVIOUS_X (1 / posterior_mean_coef1, posterior_mean_coef2 / posterior_mean_coef1), :398-405."""
        key = ("xstart", self.model_mean_type, str(device))
        if key not in self._coef_cache:
            f32 = lambda a: th.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).float()   # noqa: E731
            c = th.zeros(self.num_timesteps, 8, dtype=th.float32)
            if self.model_mean_type == ModelMeanType.EPSILON:
                c[:, 0], c[:, 1] = f32(self.sqrt_recip_alphas_cumprod), f32(self.sqrt_recipm1_alphas_cumprod)
            else:
                c[:, 0] = f32(1.0 / self.posterior_mean_coef1)
                c[:, 1] = f32(self.posterior_mean_coef2 / self.posterior_mean_coef1)
            self._coef_cache[key] = c.to(device)
        return self._coef_cache[key]

    def _identity_mean_table(self, device):
        """The ancestral coefficient rows with mean = 1 * x0-slot + 0 * x_t: PREVIOUS_X takes the denoiser output itself
        as the posterior mean (:361), and 1 * m + 0 * x is m bit for bit."""
        key = ("identity", str(device))
        if key not in self._coef_cache:
            c = self.coef_table(GDX_SAMPLER_P, device).clone()
            c[:, 0], c[:, 1] = 1.0, 0.0
            self._coef_cache[key] = c
        return self._coef_cache[key]

    def coef_table(self, kind, device, eta=0.0):
        """[num_timesteps, 8] fp32 rows consumed by gdx_sampler_update.  Every entry is rounded
        exactly like the reference: fp64 table -> .float() (gaussian_diffusion.py:1595-1608), then
        fp32 torch ops in the reference's order."""
        key = (kind, str(device), eta if eta == "reverse" else float(eta))
        if key in self._coef_cache:
            return self._coef_cache[key]
        self._check_supported()
        f32 = lambda a: th.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).float()   # noqa: E731
        n = self.num_timesteps
        nz = (th.arange(n) != 0).float()
        c = th.zeros(n, 8, dtype=th.float32)
        if kind == GDX_SAMPLER_P:
            _, logvar = self._model_variance_tables()
            c[:, 0] = f32(self.posterior_mean_coef1)
            c[:, 1] = f32(self.posterior_mean_coef2)
            c[:, 2] = nz * th.exp(0.5 * f32(logvar))
            c[:, 3] = f32(self._model_variance_tables()[0])     # model variance, used by condition_mean
        else:
            ab, abp = f32(self.alphas_cumprod), f32(self.alphas_cumprod_prev)
            sigma = (0.0 if eta == "reverse" else eta) * th.sqrt((1 - abp) / (1 - ab)) * th.sqrt(1 - ab / abp)
            c[:, 0] = f32(self.sqrt_recip_alphas_cumprod)
            c[:, 1] = f32(self.sqrt_recipm1_alphas_cumprod)
            c[:, 2] = th.sqrt(abp)
            c[:, 3] = th.sqrt(1 - abp - sigma ** 2)
            c[:, 4] = nz * sigma
        if kind == GDX_SAMPLER_DDIM and eta == "reverse":
            abn = f32(self.alphas_cumprod_next)          # ddim_reverse_sample (:841-877): same kernel, next-alpha rows
            c[:, 2] = th.sqrt(abn)
            c[:, 3] = th.sqrt(1 - abn)
            c[:, 4] = 0.0
        c[:, 5] = f32(self.sqrt_alphas_cumprod)
        c[:, 6] = f32(self.sqrt_one_minus_alphas_cumprod)
        c[:, 7] = nz                                      # (t != 0), used by the PLMS update
        c = c.to(device)
        self._coef_cache[key] = c
        return c

    # ------------------------------------------------------------------ forward process
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = th.randn_like(x_start)
        assert noise.shape == x_start.shape
        E.require_device(x_start, "x_start")
        coef = self.coef_table(GDX_SAMPLER_P, x_start.device)
        tv = E.require_device(t, "t").reshape(-1).to(th.int64).contiguous()
        return E.q_sample_t(E.f32c(x_start, "x_start"), E.f32c(noise, "noise"), coef, tv)

    def _expand(self, table, t, like):
        """_extract_into_tensor (reference :1595-1608): fp64 table -> gather -> .float() -> expanded view of `like`'s shape."""
        return th.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(like.device)[t].float().view(
            -1, *([1] * (like.dim() - 1))).expand(like.shape)

    def q_mean_variance(self, x_start, t):
        """q(x_t | x_0) (reference :216-231): (mean, variance, log_variance), all of x_start's shape.  The mean is the q_sample
        kernel with zero noise (a * x_start + b * 0); the two variance entries are per-sample table values."""
        E.require_device(x_start, "x_start")
        xs = E.f32c(x_start, "x_start")
        tv = E.require_device(t, "t").reshape(-1).to(th.int64).contiguous()
        mean = E.q_sample_t(xs, th.zeros_like(xs), self.coef_table(GDX_SAMPLER_P, xs.device), tv)
        return mean, self._expand(1.0 - self.alphas_cumprod, tv, xs), self._expand(self.log_one_minus_alphas_cumprod, tv, xs)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """q(x_{t-1} | x_t, x_0) (reference :253-275): (posterior_mean, posterior_variance, posterior_log_variance_clipped).
        The mean is the fused update kernel's coef1 * x_start + coef2 * x_t with zero noise weight."""
        assert x_start.shape == x_t.shape
        E.require_device(x_t, "x_t")
        xt, xs = E.f32c(x_t, "x_t"), E.f32c(x_start, "x_start")
        tv = E.require_device(t, "t").reshape(-1).to(th.int64).contiguous()
        mean = th.empty_like(xt)
        E.sampler_update(GDX_SAMPLER_P, self._posterior_table(xt.device), xt, xs, mean, t=tv, noise=th.zeros_like(xt))
        var, logvar = self._expand(self.posterior_variance, tv, xt), self._expand(self.posterior_log_variance_clipped, tv, xt)
        assert mean.shape[0] == var.shape[0] == logvar.shape[0] == x_start.shape[0]
        return mean, var, logvar

    def _posterior_table(self, device):
        """Ancestral coefficient rows with the POSTERIOR variance whatever model_var_type says (q_posterior_mean_variance is a
        property of the forward process; only columns 0 / 1 are used with zero noise)."""
        key = ("posterior", str(device))
        if key not in self._coef_cache:
            f32 = lambda a: th.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).float()   # noqa: E731
            c = th.zeros(self.num_timesteps, 8, dtype=th.float32)
            c[:, 0], c[:, 1] = f32(self.posterior_mean_coef1), f32(self.posterior_mean_coef2)
            self._coef_cache[key] = c.to(device)
        return self._coef_cache[key]

    def condition_mean(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """Sohl-Dickstein conditioning (reference :418-433): p_mean_var["mean"] + p_mean_var["variance"] * cond_fn(x, t, ...).
        The product / sum run in the fused update kernel (identity mean rows: 1 * mean + 0 * x, then + variance * gradient,
        zero noise weight).  The variance is this diffusion's model variance at t -- what p_mean_variance put into the dict."""
        grad = E.f32c(self._call_cond_fn(cond_fn, x, t, model_kwargs or {}), "cond_fn gradient")
        mean = E.f32c(p_mean_var["mean"], "p_mean_var['mean']")
        assert grad.shape == mean.shape == x.shape
        tv = E.require_device(t, "t").reshape(-1).to(th.int64).contiguous()
        out = th.empty_like(mean)
        E.sampler_update(GDX_SAMPLER_P, self._identity_mean_table(mean.device), E.f32c(x, "x"), mean, out, t=tv,
                         noise=th.zeros_like(mean), cond_grad=grad)
        return out

    def condition_score(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """Song et al. conditioning (reference :448-472): eps <- eps - sqrt(1 - alpha_bar) * cond_fn(x, t, ...), pred_xstart from
        it (gdx_plms_update kind 7), mean = posterior mean of the new pred_xstart.  Returns a copy of the dict."""
        xc = E.f32c(x, "x")
        grad = E.f32c(self._call_cond_fn(cond_fn, x, t, model_kwargs or {}), "cond_fn gradient")
        tv = E.require_device(t, "t").reshape(-1).to(th.int64).contiguous()
        out = dict(p_mean_var)
        out["pred_xstart"] = E.plms_update(7, self.coef_table(GDX_SAMPLER_DDIM, xc.device), tv, xc,
                                           E.f32c(p_mean_var["pred_xstart"], "pred_xstart"),
                                           eps=(grad, self._cond_coef(xc.device)))
        out["mean"], _, _ = self.q_posterior_mean_variance(x_start=out["pred_xstart"], x_t=xc, t=tv)
        return out

    # ------------------------------------------------------------------ one reverse step
    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def _call_model(self, model, x, t, model_kwargs):
        return model(x, self._scale_timesteps(t), **model_kwargs)

    def _cond_coef(self, device):
        """(1 - alpha_bar).sqrt() in fp32, the factor of condition_score (reference :463-466)."""
        key = ("cond", str(device))
        if key not in self._coef_cache:
            ab = th.from_numpy(np.ascontiguousarray(self.alphas_cumprod, dtype=np.float64)).float()
            self._coef_cache[key] = (1 - ab).sqrt().to(device)
        return self._coef_cache[key]

    def _call_cond_fn(self, cond_fn, x, t, model_kwargs):
        """cond_fn(x, t, **model_kwargs) -> gradient of log p(y | x) (reference condition_mean / condition_score)."""
        return cond_fn(x, self._scale_timesteps(t), **model_kwargs)

    def _step(self, kind, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=0.0, const_noise=False,
              noise=None):
        if model_kwargs is None:
            model_kwargs = {}
        self._check_supported()
        B, C = x.shape[:2]
        assert t.shape == (B,)
        model_output = self._call_model(model, x, t, model_kwargs)
        y = model_kwargs["y"]                                             # KeyError like the reference (:307)
        mask = motion = None
        if "inpainting_mask" in y.keys() and "inpainted_motion" in y.keys():
            mask, motion = y["inpainting_mask"], y["inpainted_motion"]
            assert model_output.shape == mask.shape == motion.shape
        x0 = E.f32c(model_output, "model output")
        xc = E.f32c(x, "x")
        tt = t.to(th.int64).contiguous()
        mean_type = self.model_mean_type
        raw = None
        if mean_type != ModelMeanType.START_X:
            # EPSILON / PREVIOUS_X (reference :357-372): the x0 prediction is an element-wise function of the output and x_t
            # (gdx_plms_update kind 8); everything downstream (process_xstart, posterior mean, DDIM / PLMS) then sees x0
            # exactly as it does for START_X.  The inpainting blend supports START_X only, as in the reference (:309).
            assert mask is None, 'This feature supports only X_start pred for mow!'
            raw = x0
            x0 = (E.plms_update(8, self._xstart_table(x.device), tt, xc, raw) if mean_type == ModelMeanType.EPSILON
                  else E.plms_update(8, self._xstart_table(x.device), tt, raw, xc))
        if denoised_fn is not None:
            # rare path (every reference caller passes denoised_fn=None): the reference applies the inpainting blend, then
            # denoised_fn, then the clamp (:307-311, :349-355).  The blend runs in the update kernel (its pred_xstart output,
            # zero noise weight irrelevant), the user's callable on the result, the clamp in the update proper.
            if mask is not None:
                blended = th.empty_like(xc)
                E.sampler_update(kind, self.coef_table(kind, x.device, eta), xc, x0, th.empty_like(xc), t=tt,
                                 inpaint_mask=mask.contiguous(), inpaint_motion=E.f32c(motion, "inpainted_motion"),
                                 noise=th.zeros_like(xc), pred_xstart=blended)
                x0, mask, motion = blended, None, None
            x0 = E.f32c(denoised_fn(x0), "denoised_fn output")
        assert x0.shape == x.shape
        if noise is None:
            noise = th.randn_like(x)                                      # drawn even at t == 0 / eta == 0
        if const_noise:
            noise = noise[[0]].contiguous()
        out = th.empty_like(xc)
        pred = th.empty_like(xc)
        grad = gcoef = None
        if cond_fn is not None:
            # condition_mean (ancestral) / condition_score (DDIM), reference :418-494: the gradient is the user's callable,
            # its application to the mean / eps happens inside the fused update kernel
            if eta == "reverse":
                raise NotImplementedError("ddim_reverse_sample takes no cond_fn")
            grad = E.f32c(self._call_cond_fn(cond_fn, x, t, model_kwargs), "cond_fn gradient")
            assert grad.shape == x.shape
            gcoef = self._cond_coef(x.device) if kind == GDX_SAMPLER_DDIM else None
        if mean_type == ModelMeanType.PREVIOUS_X and kind == GDX_SAMPLER_P:
            # the posterior mean IS the output (:361); pred_xstart is only reported (and clamped by the same kernel's
            # pred_xstart output in a pass of its own, zero noise)
            if clip_denoised:
                E.sampler_update(kind, self.coef_table(kind, x.device, eta), xc, x0, th.empty_like(xc), t=tt,
                                 noise=th.zeros_like(xc), pred_xstart=pred, clip_denoised=True)
            else:
                pred = x0
            E.sampler_update(kind, self._identity_mean_table(x.device), xc, raw, out, t=tt, noise=E.f32c(noise, "noise"),
                             const_noise=const_noise, pred_xstart=th.empty_like(xc), cond_grad=grad, cond_coef=gcoef,
                             clip_denoised=False)
            return {"sample": out, "pred_xstart": pred}
        E.sampler_update(kind, self.coef_table(kind, x.device, eta), xc, x0, out, t=tt,
                         inpaint_mask=mask.contiguous() if mask is not None else None,
                         inpaint_motion=E.f32c(motion, "inpainted_motion") if motion is not None else None,
                         noise=E.f32c(noise, "noise"), const_noise=const_noise, pred_xstart=pred, cond_grad=grad,
                         cond_coef=gcoef, clip_denoised=clip_denoised)
        return {"sample": out, "pred_xstart": pred}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                 const_noise=False):
        return self._step(GDX_SAMPLER_P, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                          const_noise=const_noise)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    eta=0.0):
        return self._step(GDX_SAMPLER_DDIM, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=eta)

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, eta=0.0):
        """Sample x_{t+1} with the DDIM reverse ODE (reference :841-877); the fused update kernel with the
        next-alpha coefficient rows and zero noise weight."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        return self._step(GDX_SAMPLER_DDIM, model, x, t, clip_denoised, denoised_fn, None, model_kwargs, eta="reverse",
                          noise=th.zeros_like(x))

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """Dict API of the reference (:277-388).  mean = update with zero noise; the variance
        entries are per-sample constants expanded to x's shape."""
        zero = th.zeros_like(x)
        r = self._step(GDX_SAMPLER_P, model, x, t, clip_denoised, denoised_fn, None, model_kwargs, noise=zero)
        var, logvar = self._model_variance_tables()
        ex = lambda a: th.from_numpy(a).to(x.device)[t].float().view(-1, *([1] * (x.dim() - 1))).expand(x.shape)  # noqa: E731
        return {"mean": r["sample"], "variance": ex(var), "log_variance": ex(logvar), "pred_xstart": r["pred_xstart"]}

    # ------------------------------------------------------------------ loops
    def _prepare_loop(self, model, shape, noise, device, skip_timesteps, init_image, rng, philox_seed, sample_offset,
                      noise_tape):
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        if noise is not None:
            img = noise
        elif noise_tape is not None:
            img = noise_tape[0]
        elif rng == "philox":
            img = E.randn(tuple(shape), device, philox_seed, sample_offset, 0)
        else:
            img = th.randn(*shape, device=device)
        if skip_timesteps and init_image is None:
            init_image = th.zeros_like(img)
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            my_t = th.ones([shape[0]], device=device, dtype=th.long) * indices[0]
            img = self.q_sample(init_image, my_t, img)
        return device, img, indices

    def _loop(self, kind, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress,
              eta, skip_timesteps, init_image, randomize_class, cond_fn_with_grad, const_noise, rng, philox_seed,
              sample_offset, noise_tape, dump_steps=None, every_step=False):
        """every_step: the caller consumes each step's dict (the *_progressive generators, or `fused=False`: the
        step-wise callable protocol model(x, t, **kw) + one gdx_sampler_update per step); otherwise only the final state
        is needed and the whole loop runs inside libgdx."""
        if cond_fn_with_grad or randomize_class:
            raise NotImplementedError("cond_fn_with_grad / randomize_class are outside the sampling hot path")
        if rng not in ("torch", "philox"):
            raise ValueError(f"rng must be 'torch' or 'philox', got {rng!r}")
        device, img, indices = self._prepare_loop(model, shape, noise, device, skip_timesteps, init_image, rng,
                                                  philox_seed, sample_offset, noise_tape)
        fused = (not every_step and _is_native(model) and denoised_fn is None and cond_fn is None
                 and self.model_mean_type == ModelMeanType.START_X)
        if fused:
            yield from self._fused_loop(kind, model, img, indices, model_kwargs, eta, const_noise, rng, philox_seed,
                                        sample_offset, noise_tape, dump_steps, progress, clip_denoised)
            return
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for k, i in enumerate(indices):
            t = th.full((shape[0],), i, device=device, dtype=th.long)
            z = None
            if noise_tape is not None:
                z = noise_tape[1 + k]
            elif rng == "philox":
                z = E.randn(tuple(shape), device, philox_seed, 0 if const_noise else sample_offset, k + 1)
            with th.no_grad():
                out = self._step(kind, model, img, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=eta,
                                 const_noise=const_noise, noise=z)
            yield out
            img = out["sample"]

    def _fused_loop(self, kind, model, img, indices, model_kwargs, eta, const_noise, rng, philox_seed, sample_offset,
                    noise_tape, dump_steps, progress, clip_denoised=False):
        """Whole loop inside libgdx (gdx_sample_loop); yields only the final state.  Noise: a recorded tape, in-kernel
        Philox, or torch's generator -- then one `normal_()` per step in the reference's order (:532: randn_like(x) is
        empty_like(x).normal_()), drawn NOISE_BLOCK steps ahead into a tape the update kernel reads; the loop is issued
        block by block with no host synchronisation in between (with `progress` one per block, to report it)."""
        from ..model.cfg_sampler import ClassifierFreeSampleModel
        self._check_supported()
        y = model_kwargs["y"]
        inner = model.model if isinstance(model, ClassifierFreeSampleModel) else model
        x = E.f32c(img, "x_T").clone()
        B, J, F, T = x.shape
        inner._check_inputs(x, y)
        if hasattr(inner, "cl_head") and T % 10 != 0:
            from ..model.mdm import _window_error
            raise _window_error(T, 10)
        eng = inner._get_engine(x.device)
        eng.prepare(B, T)
        eng.set_condition(y["seed"], y["mfcc"], cache=False)
        if isinstance(model, ClassifierFreeSampleModel):
            mode, scale = GDX_CFG, E.f32c(y["scale"].reshape(-1), "y['scale']")
        else:
            mode, scale = (GDX_UNCOND if y.get("uncond", False) else GDX_COND), None
        mask = motion = None
        if "inpainting_mask" in y and "inpainted_motion" in y:
            mask = E.require_device(y["inpainting_mask"], "inpainting_mask").to(th.bool).contiguous()
            motion = E.f32c(y["inpainted_motion"], "inpainted_motion")
            assert mask.shape == motion.shape == x.shape
        n = len(indices)
        tape = None
        if noise_tape is not None:
            tape = E.f32c(noise_tape[1:1 + n], "noise_tape")
            if const_noise:
                tape = tape[:, :1].contiguous()
        tmap = self._timestep_map()
        dump = None
        if dump_steps is not None:
            dump_steps = sorted({int(s) for s in dump_steps if 0 <= int(s) < n})     # `i in dump_steps`: each step once
            dump = th.empty(len(dump_steps), *x.shape, device=x.device, dtype=th.float32)
        if self.rescale_timesteps:
            raise NotImplementedError("rescale_timesteps=True is not used by the reference's sampler configuration")
        draw = tape is None and rng == "torch"
        block = noise_block_steps(n, (1 if const_noise else B) * J * F * T) if draw else (min(n, NOISE_BLOCK) if progress else n)
        bar = None
        if progress:
            from tqdm.auto import tqdm
            bar = tqdm(total=n)
        coef = self.coef_table(kind, x.device, eta)
        buf = th.empty((block, 1 if const_noise else B, J, F, T), device=x.device, dtype=th.float32) if draw else None
        full = th.empty_like(x) if draw and const_noise else None
        k = 0
        while k < n:
            nb = min(block, n - k)
            if draw:
                for j in range(nb):
                    if const_noise:
                        buf[j].copy_(full.normal_()[:1])      # reference :534-535: the full draw, sample 0 kept
                    else:
                        buf[j].normal_()
                blk = buf
            else:
                blk = tape[k:] if tape is not None else None
            eng.sample_loop(x, kind, mode, coef, tmap, indices[k], scale=scale, inpaint_mask=mask,
                            inpaint_motion=motion, noise_tape=blk, const_noise=const_noise, philox_seed=philox_seed,
                            sample_offset=sample_offset, dump=dump, dump_steps=dump_steps, run_steps=nb, k_base=k,
                            clip_denoised=clip_denoised)
            k += nb
            if bar is not None:
                th.cuda.current_stream(x.device).synchronize()
                bar.update(nb)
        if bar is not None:
            bar.close()
        yield {"sample": x, "pred_xstart": None, "dump": dump, "fused": True}

    def _timestep_map(self):
        return list(range(self.num_timesteps))

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                      randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                      rng="torch", philox_seed=0, sample_offset=0, noise_tape=None, fused=True):
        final = None
        dump = [] if dump_steps is not None else None
        for i, sample in enumerate(self._loop(GDX_SAMPLER_P, model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                                              model_kwargs, device, progress, 0.0, skip_timesteps, init_image,
                                              randomize_class, cond_fn_with_grad, const_noise, rng, philox_seed,
                                              sample_offset, noise_tape, dump_steps, every_step=not fused)):
            if sample.get("fused"):
                if dump_steps is not None:
                    return [d.clone() for d in sample["dump"]]
                return sample["sample"]
            if dump_steps is not None and i in dump_steps:
                dump.append(sample["sample"].clone())
            final = sample
        if dump_steps is not None:
            return dump
        return final["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                                  randomize_class=False, cond_fn_with_grad=False, const_noise=False):
        yield from self._loop(GDX_SAMPLER_P, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                              device, progress, 0.0, skip_timesteps, init_image, randomize_class, cond_fn_with_grad,
                              const_noise, "torch", 0, 0, None, every_step=True)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0, init_image=None,
                         randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                         rng="torch", philox_seed=0, sample_offset=0, noise_tape=None, fused=True):
        if dump_steps is not None:
            raise NotImplementedError()                                   # reference :903-904
        if const_noise == True:  # noqa: E712
            raise NotImplementedError()                                   # reference :905-906
        final = None
        for sample in self._loop(GDX_SAMPLER_DDIM, model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                                 model_kwargs, device, progress, eta, skip_timesteps, init_image, randomize_class,
                                 cond_fn_with_grad, False, rng, philox_seed, sample_offset, noise_tape,
                                 every_step=not fused):
            final = sample
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0,
                                     skip_timesteps=0, init_image=None, randomize_class=False,
                                     cond_fn_with_grad=False):
        yield from self._loop(GDX_SAMPLER_DDIM, model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                              model_kwargs, device, progress, eta, skip_timesteps, init_image, randomize_class,
                              cond_fn_with_grad, False, "torch", 0, 0, None, every_step=True)

    # ------------------------------------------------------------------ explicitly out of scope
    def masked_l2(self, a, b, mask):
        """reference :201-213; a, b [B,J,1,T], mask bool [B,1,1,T] -> [B]."""
        return E.masked_l2(E.f32c(a, "a"), E.f32c(b, "b"), mask)

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None, dataset=None):
        """FORWARD half of the reference's training_losses (:1227-1352) in its configured mode (LossType.MSE, START_X,
        fixed variance, lambda_vel = lambda_rcxyz = lambda_fc = 0): x_t = q_sample(x_start, t, noise), the model's x0
        prediction, terms['rot_mse'] = masked_l2(x_start, prediction, y['mask']), terms['loss'] = rot_mse.  The values are
        those the reference would log; there is no autograd graph behind them (training is out of scope)."""
        from . import gaussian_diffusion as _gd
        if self.loss_type not in (_gd.LossType.MSE, _gd.LossType.RESCALED_MSE):
            raise NotImplementedError(self.loss_type)
        self._check_supported()
        for lam in ("lambda_vel", "lambda_rcxyz", "lambda_fc"):
            if getattr(self, lam, 0.0):
                raise NotImplementedError(f"{lam} > 0 needs the xyz / velocity terms, which are not on the path")
        mask = model_kwargs["y"]["mask"]                       # KeyError / TypeError like the reference (:1243)
        if noise is None:
            noise = th.randn_like(x_start)
        x_t = self.q_sample(x_start, t, noise=noise)
        with th.no_grad():
            model_output = self._call_model(model, x_t, t, model_kwargs)
        assert model_output.shape == x_start.shape
        terms = {"rot_mse": self.masked_l2(x_start, model_output, mask)}
        terms["loss"] = terms["rot_mse"]
        return terms

    # ------------------------------------------------------------------ PLMS (reference :995-1190)
    def _pred_xstart(self, model, x, t, clip_denoised, denoised_fn, model_kwargs):
        """p_mean_variance's pred_xstart (model output after CFG / inpainting / clipping)."""
        return self._step(GDX_SAMPLER_P, model, x, t, clip_denoised, denoised_fn, None, model_kwargs,
                          noise=th.zeros_like(x))["pred_xstart"]

    def plms_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    cond_fn_with_grad=False, order=2, old_out=None):
        """Pseudo linear multistep step (reference :995-1079): eps from the x0 prediction, pseudo improved Euler on
        the first step (one extra model call at t-1), Adams-Bashforth of order <= `order` afterwards.  The
        element-wise arithmetic runs in gdx_plms_update with the reference's op order."""
        if not int(order) or not 1 <= order <= 4:
            raise ValueError('order is invalid (should be int from 1-4).')
        if cond_fn_with_grad:
            raise NotImplementedError("cond_fn_with_grad needs autograd through the denoiser (outside the hot path)")
        coef = self.coef_table(GDX_SAMPLER_DDIM, x.device, 0.0)
        xc = E.f32c(x, "x")
        tt = t.to(th.int64).contiguous()

        def get_model_output(xx, ts):
            """(eps, pred_xstart used downstream, original pred_xstart), reference get_model_output :1015-1041."""
            orig = self._pred_xstart(model, xx, ts, clip_denoised, denoised_fn, model_kwargs)
            used = orig
            if cond_fn is not None:                       # condition_score (:452-472)
                grad = E.f32c(self._call_cond_fn(cond_fn, xx, ts, model_kwargs), "cond_fn gradient")
                used = E.plms_update(7, coef, ts, xx, orig, eps=[grad, self._cond_coef(xx.device)])
            return E.plms_update(0, coef, ts, xx, used), used, orig
        eps, x0, x0_orig = get_model_output(xc, tt)
        if order > 1 and old_out is None:
            old_eps = [eps]
            mean_pred = E.plms_update(6, coef, tt, None, x0, eps=[eps])
            t2 = (tt - 1) % self.num_timesteps            # t - 1 (a negative index wraps in the reference's table gather)
            eps_2, _, _ = get_model_output(mean_pred, t2)
            sample = E.plms_update(5, coef, tt, xc, x0, eps=[eps, eps_2])
        else:
            old_eps = old_out["old_eps"]                  # TypeError on the first step with order == 1, like the reference
            old_eps.append(eps)
            cur = min(order, len(old_eps))
            sample = E.plms_update(cur, coef, tt, xc, x0, eps=old_eps[::-1][:cur])
        if len(old_eps) >= order:
            old_eps.pop(0)
        return {"sample": sample, "pred_xstart": x0_orig, "old_eps": old_eps}

    def plms_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                         randomize_class=False, cond_fn_with_grad=False, order=2):
        final = None
        for sample in self.plms_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                        denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                        model_kwargs=model_kwargs, device=device, progress=progress,
                                                        skip_timesteps=skip_timesteps, init_image=init_image,
                                                        randomize_class=randomize_class,
                                                        cond_fn_with_grad=cond_fn_with_grad, order=order):
            final = sample
        return final["sample"]

    def plms_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, skip_timesteps=0,
                                     init_image=None, randomize_class=False, cond_fn_with_grad=False, order=2):
        if randomize_class:
            raise NotImplementedError("randomize_class is outside the sampling hot path")
        device, img, indices = self._prepare_loop(model, shape, noise, device, skip_timesteps, init_image, "torch", 0,
                                                  0, None)
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        old_out = None
        for i in indices:
            t = th.full((shape[0],), i, device=device, dtype=th.long)
            with th.no_grad():
                out = self.plms_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                       cond_fn=cond_fn, model_kwargs=model_kwargs,
                                       cond_fn_with_grad=cond_fn_with_grad, order=order, old_out=old_out)
            yield out
            old_out = out
            img = out["sample"]


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """Kept for API parity (reference :1595-1608)."""
    res = th.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)
