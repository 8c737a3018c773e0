"""Oracle: sampler arithmetic and loops, CPU fp32 (test infrastructure only).

Restates, for the configured mode START_X + FIXED_SMALL (utils/model_util.py:37-72) and, in `mean_type_step`, for the
EPSILON / PREVIOUS_X readings of the denoiser output and FIXED_LARGE variance (p_mean_variance :316-372),
reference `diffusion/gaussian_diffusion.py`:
  q_sample :233-251, q_posterior_mean_variance :253-275, p_mean_variance :277-388
  (incl. the inpainting blend :307-311), _predict_eps_from_xstart :407-411,
  p_sample :496-548, p_sample_loop{,_progressive} :598-730, ddim_sample :732-782,
  ddim_sample_loop{,_progressive} :879-993, _extract_into_tensor :1595-1608,
and `diffusion/respace.py:117-129` (_WrappedModel timestep mapping).

The reference draws its Gaussians from torch's global generator; the oracle replays
an explicit *noise tape* ``[x_T, z_{N-1}, ..., z_0]`` instead so CPU/GPU runs see the
same numbers.  Pinned by tests/golden/loops_*.npz.
"""
import torch


def extract(arr, t):
    """_extract_into_tensor: fp64 table -> gather -> .float() -> [B,1,1,1]."""
    return torch.from_numpy(arr)[t].float().view(-1, 1, 1, 1)


def q_sample(tab, x_start, t, noise):
    return extract(tab.sqrt_alphas_cumprod, t) * x_start + extract(tab.sqrt_one_minus_alphas_cumprod, t) * noise


def inpaint(x0, y):
    if "inpainting_mask" in y and "inpainted_motion" in y:
        m = y["inpainting_mask"]
        return (x0 * ~m) + (y["inpainted_motion"] * m)
    return x0


def p_sample_step(tab, x0, x, t, noise):
    """Ancestral update given the model's x0 prediction (FIXED_SMALL variance)."""
    mean = extract(tab.posterior_mean_coef1, t) * x0 + extract(tab.posterior_mean_coef2, t) * x
    log_var = extract(tab.posterior_log_variance_clipped, t)
    nonzero = (t != 0).float().view(-1, 1, 1, 1)
    return mean + nonzero * torch.exp(0.5 * log_var) * noise


def ddim_step(tab, x0, x, t, noise, eta=0.0):
    eps = (extract(tab.sqrt_recip_alphas_cumprod, t) * x - x0) / extract(tab.sqrt_recipm1_alphas_cumprod, t)
    ab = extract(tab.alphas_cumprod, t)
    ab_prev = extract(tab.alphas_cumprod_prev, t)
    sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
    mean = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    nonzero = (t != 0).float().view(-1, 1, 1, 1)
    return mean + nonzero * sigma * noise


def ddim_reverse_step(tab, x0, x, t):
    """reference gaussian_diffusion.py:841-877 (deterministic x_t -> x_{t+1})."""
    eps = (extract(tab.sqrt_recip_alphas_cumprod, t) * x - x0) / extract(tab.sqrt_recipm1_alphas_cumprod, t)
    ab_next = extract(tab.alphas_cumprod_next, t)
    return x0 * torch.sqrt(ab_next) + torch.sqrt(1 - ab_next) * eps


def process_xstart(x0, clip_denoised=False, denoised_fn=None):
    """reference gaussian_diffusion.py:349-355 (applied to the model output AFTER the inpainting blend :307-311)."""
    if denoised_fn is not None:
        x0 = denoised_fn(x0)
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    return x0


def predict_xstart_from_xprev(tab, xprev, x, t):
    """_predict_xstart_from_xprev :398-405 (tables divided in fp64, then gathered and rounded like every other)."""
    return extract(1.0 / tab.posterior_mean_coef1, t) * xprev - extract(tab.posterior_mean_coef2 / tab.posterior_mean_coef1, t) * x


def model_log_variance(tab, t, var_large=False):
    """FIXED_SMALL / FIXED_LARGE rows of p_mean_variance (:332-352)."""
    import numpy as np
    if var_large:
        return extract(np.log(np.append(tab.posterior_variance[1], tab.betas[1:])), t)
    return extract(tab.posterior_log_variance_clipped, t)


def mean_type_step(tab, out, x, t, noise, y, kind, mean_type, eta=0.0, var_large=False, clip_denoised=False,
                   denoised_fn=None):
    """One p_sample / ddim_sample step for a denoiser output read as START_X, EPSILON or PREVIOUS_X (p_mean_variance
    :357-372).  The inpainting blend asserts START_X in the reference (:309)."""
    if "inpainting_mask" in y and "inpainted_motion" in y:
        assert mean_type == "start_x", 'This feature supports only X_start pred for mow!'
        out = inpaint(out, y)
    if mean_type == "previous_x":
        x0 = process_xstart(predict_xstart_from_xprev(tab, out, x, t), clip_denoised, denoised_fn)
        mean = out
    else:
        x0 = process_xstart(out if mean_type == "start_x" else predict_xstart(tab, out, x, t), clip_denoised, denoised_fn)
        mean = extract(tab.posterior_mean_coef1, t) * x0 + extract(tab.posterior_mean_coef2, t) * x
    if kind == "ddim":
        return ddim_step(tab, x0, x, t, noise, eta), x0
    nonzero = (t != 0).float().view(-1, 1, 1, 1)
    return mean + nonzero * torch.exp(0.5 * model_log_variance(tab, t, var_large)) * noise, x0


def sample_loop(model_fn, tab, tmap, shape, tape, y, kind="p", eta=0.0, skip_timesteps=0,
                init_image=None, const_noise=False, dump_steps=None, clip_denoised=False, denoised_fn=None,
                mean_type="start_x", var_large=False):
    """Drive `model_fn(x, mapped_t, y) -> x0` through the whole reverse process.

    tape: list/tensor of N+1 noise tensors, tape[0] = x_T, tape[1+k] = z of the k-th
    executed step (reference draws in exactly this order).
    Returns the final sample (or the list of dumped steps, ancestral only).
    """
    B = shape[0]
    img = tape[0]
    if skip_timesteps and init_image is None:
        init_image = torch.zeros_like(img)
    indices = list(range(tab.num_timesteps - skip_timesteps))[::-1]
    if init_image is not None:
        my_t = torch.ones(B, dtype=torch.long) * indices[0]
        img = q_sample(tab, init_image, my_t, img)
    map_tensor = torch.tensor(tmap, dtype=torch.long)
    dump = []
    for k, i in enumerate(indices):
        t = torch.tensor([i] * B)
        if mean_type != "start_x" or var_large:
            img, _ = mean_type_step(tab, model_fn(img, map_tensor[t], y), img, t, tape[1 + k], y, kind, mean_type, eta,
                                    var_large, clip_denoised, denoised_fn)
            continue
        x0 = process_xstart(inpaint(model_fn(img, map_tensor[t], y), y), clip_denoised, denoised_fn)
        z = tape[1 + k]
        if kind == "p":
            if const_noise:
                z = z[[0]].repeat(B, 1, 1, 1)
            img = p_sample_step(tab, x0, img, t, z)
        else:
            img = ddim_step(tab, x0, img, t, z, eta)
        if dump_steps is not None and k in dump_steps:
            dump.append(img.clone())
    return dump if dump_steps is not None else img


# ---------------------------------------------------------------------------------------------
# PLMS (reference gaussian_diffusion.py:995-1190), for the configured mode START_X.
def predict_eps(tab, x0, x, t):
    """_predict_eps_from_xstart :407-411."""
    return (extract(tab.sqrt_recip_alphas_cumprod, t) * x - x0) / extract(tab.sqrt_recipm1_alphas_cumprod, t)


def predict_xstart(tab, eps, x, t):
    """_predict_xstart_from_eps :390-396."""
    return extract(tab.sqrt_recip_alphas_cumprod, t) * x - extract(tab.sqrt_recipm1_alphas_cumprod, t) * eps


def plms_step(x0_fn, tab, x, t, order, old_eps):
    """plms_sample :995-1079.  x0_fn(x, t) -> x0 prediction after the inpainting blend (p_mean_variance) and, when a
    cond_fn is used, after condition_score (pass `lambda x, t: cond_xstart(tab, raw_x0_fn(x, t), x, t, grad_fn(x, t))`).
    old_eps: list carried between steps (None on the first step).  Returns (sample, pred_xstart, old_eps)."""
    if not int(order) or not 1 <= order <= 4:
        raise ValueError('order is invalid (should be int from 1-4).')
    ab_prev = extract(tab.alphas_cumprod_prev, t)
    x0 = x0_fn(x, t)
    eps = predict_eps(tab, x0, x, t)
    if order > 1 and old_eps is None:
        old_eps = [eps]                                                       # pseudo improved Euler
        mean_pred = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev) * eps
        eps_2 = predict_eps(tab, x0_fn(mean_pred, t - 1), mean_pred, t - 1)
        eps_prime = (eps + eps_2) / 2
    else:
        if old_eps is None:                                                   # order == 1 on the first step: the reference
            raise TypeError("'NoneType' object is not subscriptable")         # subscripts old_out = None (:1057)
        old_eps.append(eps)
        cur = min(order, len(old_eps))
        if cur == 1:
            eps_prime = old_eps[-1]
        elif cur == 2:
            eps_prime = (3 * old_eps[-1] - old_eps[-2]) / 2
        elif cur == 3:
            eps_prime = (23 * old_eps[-1] - 16 * old_eps[-2] + 5 * old_eps[-3]) / 12
        else:
            eps_prime = (55 * old_eps[-1] - 59 * old_eps[-2] + 37 * old_eps[-3] - 9 * old_eps[-4]) / 24
    pred_prime = predict_xstart(tab, eps_prime, x, t)
    mean_pred = pred_prime * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev) * eps_prime
    if len(old_eps) >= order:
        old_eps.pop(0)
    nonzero = (t != 0).float().view(-1, 1, 1, 1)
    return mean_pred * nonzero + x0 * (1 - nonzero), x0, old_eps


def plms_loop(model_fn, tab, tmap, shape, x_T, y, order=2, skip_timesteps=0, init_image=None, mean_type="start_x",
              clip_denoised=False):
    """plms_sample_loop_progressive :1121-1190 (deterministic given x_T)."""
    B = shape[0]
    img = x_T
    if skip_timesteps and init_image is None:
        init_image = torch.zeros_like(img)
    indices = list(range(tab.num_timesteps - skip_timesteps))[::-1]
    if init_image is not None:
        img = q_sample(tab, init_image, torch.ones(B, dtype=torch.long) * indices[0], img)
    map_tensor = torch.tensor(tmap, dtype=torch.long)
    x0_fn = lambda x, t: inpaint(model_fn(x, map_tensor[t], y), y)   # noqa: E731
    if mean_type != "start_x" or clip_denoised:                      # pred_xstart of p_mean_variance for the other readings
        x0_fn = lambda x, t: mean_type_step(tab, model_fn(x, map_tensor[t], y), x, t, torch.zeros_like(x), y, "p",   # noqa: E731
                                            mean_type, clip_denoised=clip_denoised)[1]
    old = None
    for i in indices:
        img, _, old = plms_step(x0_fn, tab, img, torch.tensor([i] * B), order, old)
    return img


# ---------------------------------------------------------------------------------------------
def sample_chunks(model_fn, tab, tmap, first_seed, mfccs, tapes, seed_poses, scale=None, kind="p", eta=0.0):
    """The chunked autoregressive driver, reference sample/generate.py:91-130: chunk c is one whole sampling loop whose
    y['seed'] is `first_seed` for c = 0 and afterwards the last `seed_poses` frames of chunk c-1's output (:104-107);
    y['scale'] = ones * guidance_param when guidance is on (:114-115).  tapes[c] = [x_T, z_0, ...] of chunk c.
    Pinned by tests/golden/chunks_tiny.npz (the reference's own p_sample_loop + ClassifierFreeSampleModel driven through
    these statements)."""
    outs, sample_out = [], None
    for c, (mfcc, tape) in enumerate(zip(mfccs, tapes)):
        y = {"mfcc": mfcc, "seed": first_seed if c == 0 else sample_out[..., -seed_poses:]}
        if scale is not None:
            y["scale"] = torch.ones(first_seed.shape[0]) * scale
        sample_out = sample_loop(model_fn, tab, tmap, tuple(tape[0].shape), tape, y, kind=kind, eta=eta)
        outs.append(sample_out)
    return outs


def postprocess_chunk(sample_out, mean, std):
    """Tail of the reference's chunk loop, sample/generate.py:132-146 (rot2xyz with pose_rep 'xyz' is the identity):
    inv_transform on [B, 1, T, J] (torch fp32 * numpy fp64 statistics -> fp64, then .float()), then the position /
    rotation index split and the permute to [B, n_joints, 3, T]."""
    import numpy as np
    n_joints = sample_out.shape[1] // 6
    sample = (sample_out.cpu().permute(0, 2, 3, 1) * torch.from_numpy(np.asarray(std)) + torch.from_numpy(np.asarray(mean))).float()
    idx_positions = np.asarray([[i * 6 + 3, i * 6 + 4, i * 6 + 5] for i in range(n_joints)]).flatten()
    idx_rotations = np.asarray([[i * 6, i * 6 + 1, i * 6 + 2] for i in range(n_joints)]).flatten()
    pos, rot = sample[..., idx_positions], sample[..., idx_rotations]
    pos = pos.view(pos.shape[:-1] + (-1, 3))
    pos = pos.view(-1, *pos.shape[2:]).permute(0, 2, 3, 1)
    rot = rot.view(rot.shape[:-1] + (-1, 3))
    rot = rot.view(-1, *rot.shape[2:]).permute(0, 2, 3, 1)
    return pos, rot


# ---------------------------------------------------------------------------------------------
def masked_l2(a, b, mask):
    """gaussian_diffusion.py:201-213 (mask [B,1,1,T] bool)."""
    loss = ((a - b) ** 2 * mask.float()).sum(dim=[1, 2, 3])
    n_entries = a.shape[1] * a.shape[2]
    return loss / (mask.sum(dim=[1, 2, 3]) * n_entries)


def training_losses(model_fn, tab, tmap, x_start, t, y, noise):
    """Forward half of training_losses (gaussian_diffusion.py:1227-1352) in the configured mode (LossType.MSE,
    START_X, fixed variance, lambda_vel = lambda_rcxyz = lambda_fc = 0): x_t = q_sample(x_start, t, noise),
    model_output = model(x_t, t), rot_mse = masked_l2(x_start, model_output, y['mask']); loss = rot_mse."""
    x_t = q_sample(tab, x_start, t, noise)
    out = model_fn(x_t, torch.tensor(tmap, dtype=torch.long)[t], y)
    rot = masked_l2(x_start, out, y["mask"])
    return {"rot_mse": rot, "loss": rot}


# ---------------------------------------------------------------------------------------------
# cond_fn guidance (reference gaussian_diffusion.py:418-494; respace.py:99-103 wraps cond_fn so it sees mapped timesteps)
def p_sample_step_cond(tab, x0, x, t, noise, grad):
    """p_sample :496-548 with condition_mean :418-433: mean + variance * gradient (FIXED_SMALL variance)."""
    mean = extract(tab.posterior_mean_coef1, t) * x0 + extract(tab.posterior_mean_coef2, t) * x
    mean = mean.float() + extract(tab.posterior_variance, t) * grad.float()
    nonzero = (t != 0).float().view(-1, 1, 1, 1)
    return mean + nonzero * torch.exp(0.5 * extract(tab.posterior_log_variance_clipped, t)) * noise


def ddim_step_cond(tab, x0, x, t, noise, grad, eta=0.0):
    """ddim_sample :732-782 with condition_score :452-472: eps <- eps - sqrt(1 - alpha_bar) * gradient, pred_xstart from it."""
    ab = extract(tab.alphas_cumprod, t)
    eps = predict_eps(tab, x0, x, t)
    eps = eps - (1 - ab).sqrt() * grad
    x0c = predict_xstart(tab, eps, x, t)
    return ddim_step(tab, x0c, x, t, noise, eta)


def cond_xstart(tab, x0, x, t, grad):
    """pred_xstart under condition_score (:452-472)."""
    eps = predict_eps(tab, x0, x, t) - (1 - extract(tab.alphas_cumprod, t)).sqrt() * grad
    return predict_xstart(tab, eps, x, t)


# ---------------------------------------------------------------------------------------------
# public helper methods of GaussianDiffusion (reference :216-231, :253-275, :418-433, :448-472)
def q_mean_variance(tab, x_start, t):
    shape = x_start.shape
    return (extract(tab.sqrt_alphas_cumprod, t) * x_start, extract(1.0 - tab.alphas_cumprod, t).expand(shape),
            extract(tab.log_one_minus_alphas_cumprod, t).expand(shape))


def q_posterior_mean_variance(tab, x_start, x_t, t):
    shape = x_t.shape
    mean = extract(tab.posterior_mean_coef1, t) * x_start + extract(tab.posterior_mean_coef2, t) * x_t
    return mean, extract(tab.posterior_variance, t).expand(shape), extract(tab.posterior_log_variance_clipped, t).expand(shape)


def condition_mean(mean, variance, grad):
    return mean.float() + variance * grad.float()


def condition_score(tab, pred_xstart, x, t, grad):
    """-> (pred_xstart, mean) of the dict condition_score returns."""
    x0c = cond_xstart(tab, pred_xstart, x, t, grad)
    return x0c, q_posterior_mean_variance(tab, x0c, x, t)[0]
