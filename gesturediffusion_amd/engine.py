"""Thin PyTorch-ROCm <-> libgdx.so glue: PyTorch supplies device memory and the stream,
the C ABI (include/gdx.h) does all the arithmetic.  No torch compute happens here."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import GDX_ARCH_MDM, GDX_ARCH_MDM_OLD, GDX_CFG, GDX_COND, GDX_UNCOND, GdxError  # noqa: F401

MFCC_DIM = 26


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_device(t, what):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise GdxError(f"{what} is on {t.device}: the MI355X HIP path needs device tensors "
                       f"(there is no CPU fallback in gesturediffusion_amd)")
    return t


def f32c(t, what):
    require_device(t, what)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# gdx.h GDX_DTYPE_*: "fp32" = exact fp32 MFMA everywhere (default); "fp16" / "bf16" = 16-bit GEMM / attention operands and
# activation stream, fp32 accumulate (bf16: fp32's exponent range, 8 significant bits)
COMPUTE_DTYPES = {"fp32": 0, "fp16": 1, "bf16": 2}


class Engine:
    """One libgdx handle = one model instance on one device/stream."""

    def __init__(self, arch, njoints, latent_dim, ff_size, num_layers, num_heads, seed_poses, cl_head=8, window=10,
                 compute_dtype="fp32"):
        self.lib = _lib.load()
        if compute_dtype not in COMPUTE_DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(COMPUTE_DTYPES)}, got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        self.cfg = _lib.Config(arch=arch, njoints=njoints, latent_dim=latent_dim, ff_size=ff_size,
                               num_layers=num_layers, num_heads=num_heads, seed_poses=seed_poses, mfcc_dim=MFCC_DIM,
                               cl_head=cl_head, window=window, compute_dtype=COMPUTE_DTYPES[compute_dtype])
        self.handle = C.c_void_p()
        _lib.check(self.lib.gdx_create(C.byref(self.cfg), C.byref(self.handle)), self.lib)
        self.device = None
        self.shape = None          # (B, T) the workspace is sized for
        self._cond_key = None
        self._weights_key = None
        self._keep = []            # tensors that must outlive enqueued work

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.gdx_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def set_graph_replay(self, on):
        """hipGraph replay of the step inside sample_loop (off by default; see include/gdx.h)."""
        _lib.check(self.lib.gdx_set_graph_replay(self.handle, int(bool(on))), self.lib)

    # ------------------------------------------------------------------ weights
    def set_weight(self, name, t):
        t = f32c(t, name)
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(self.lib.gdx_set_weight(self.handle, name.encode(), _ptr(t), shape, t.dim(), _stream(t.device)),
                   self.lib)
        self.device = t.device
        self._cond_key = None

    def load_tensors(self, named):
        for name, t in named.items():
            self.set_weight(name, t)
        _lib.check(self.lib.gdx_weights_ready(self.handle), self.lib)

    def export_packed(self, device):
        """The handle's packed operand layout as one bytes object (gdx_export_packed)."""
        n = C.c_int64()
        _lib.check(self.lib.gdx_packed_bytes(self.handle, C.byref(n)), self.lib)
        buf = (C.c_char * n.value)()
        _lib.check(self.lib.gdx_export_packed(self.handle, buf, n.value, _stream(device)), self.lib)
        return bytes(buf)

    def import_packed(self, blob, device):
        """Upload a blob written by export_packed; raises GdxError (handle untouched) if it was built for another
        configuration, compute dtype or shape."""
        if torch.device(device).type != "cuda":
            raise GdxError(f"packed image target is {device}: the MI355X HIP path needs a device (no CPU fallback)")
        _lib.check(self.lib.gdx_import_packed(self.handle, blob, len(blob), _stream(device)), self.lib)
        self.device = device
        self._cond_key = None

    # ------------------------------------------------------------------ shapes / conditioning
    def prepare(self, batch, frames):
        if self.shape != (batch, frames):
            _lib.check(self.lib.gdx_prepare(self.handle, batch, frames), self.lib)
            self.shape = (batch, frames)
            self._cond_key = None

    def set_condition(self, seed, mfcc, cache=True):
        """seed [B,J,1,P], mfcc [B,26,1,T].  The step-wise callable protocol model(x, t, y=...) calls this on every
        step, so the step-invariant work is skipped when the SAME tensors come back: the key is (storage pointer,
        version counter, shape, strides) and the engine keeps references to the caller's own tensors while the key is
        live -- their storage can therefore not be freed and handed to a different tensor with an equal key (a view
        such as sample_out[..., -seed_poses:] shares its base's storage and version counter).  `cache=False` (the
        fused loop: one call per 1000 steps) always recomputes."""
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype) for t in (seed, mfcc))
        if cache and key == self._cond_key:
            return
        seed_c, mfcc_c = f32c(seed, "y['seed']"), f32c(mfcc, "y['mfcc']")
        _lib.check(self.lib.gdx_set_condition(self.handle, _ptr(seed_c), _ptr(mfcc_c), _stream(seed_c.device)),
                   self.lib)
        self._keep = [seed, mfcc, seed_c, mfcc_c]
        self._cond_key = key if cache else None

    # ------------------------------------------------------------------ compute
    def forward(self, x, timesteps, mode=GDX_COND, scale=None):
        x = f32c(x, "x")
        t = require_device(timesteps, "timesteps").to(torch.int64).contiguous()
        out = torch.empty_like(x)
        sc = f32c(scale, "y['scale']") if scale is not None else None
        _lib.check(self.lib.gdx_forward(self.handle, _ptr(x), _ptr(t), mode, _ptr(sc), _ptr(out), _stream(x.device)),
                   self.lib)
        return out

    def keep_taps(self, keep=True):
        _lib.check(self.lib.gdx_set_keep_taps(self.handle, int(keep)), self.lib)
        self._cond_key = None if keep else self._cond_key

    def set_guards(self, on=True):
        """Test aid: canary zones behind every workspace buffer (gdx_set_guards)."""
        _lib.check(self.lib.gdx_set_guards(self.handle, int(bool(on))), self.lib)
        self.shape = self.shape          # the workspace was re-allocated: conditioning must be set again
        self._cond_key = None

    def check_guards(self, device):
        bad, first = C.c_int64(), C.c_int32()
        _lib.check(self.lib.gdx_check_guards(self.handle, C.byref(bad), C.byref(first), _stream(device)), self.lib)
        return bad.value, first.value

    def tap(self, which, rows, d, device):
        out = torch.empty(rows, d, device=device, dtype=torch.float32)
        _lib.check(self.lib.gdx_get_tap(self.handle, which, _ptr(out), out.numel(), _stream(device)), self.lib)
        return out

    def sample_loop(self, x, kind, mode, coef, timestep_map, first_index, scale=None, inpaint_mask=None,
                    inpaint_motion=None, noise_tape=None, const_noise=False, philox_seed=0, sample_offset=0,
                    dump=None, dump_steps=None, run_steps=0, k_base=0, clip_denoised=False):
        """x is updated in place (x_T in, final sample out).  run_steps / k_base: one block of a loop (gdx.h)."""
        tmap = np.ascontiguousarray(np.asarray(timestep_map, dtype=np.int64))
        ds = np.ascontiguousarray(np.asarray(dump_steps if dump_steps is not None else [], dtype=np.int32))
        a = _lib.LoopArgs(kind=kind, mode=mode, num_steps=len(tmap), first_index=first_index, coef=coef.data_ptr(),
                          timestep_map=tmap.ctypes.data, x=x.data_ptr(),
                          scale=scale.data_ptr() if scale is not None else None,
                          inpaint_mask=inpaint_mask.data_ptr() if inpaint_mask is not None else None,
                          inpaint_motion=inpaint_motion.data_ptr() if inpaint_motion is not None else None,
                          noise_tape=noise_tape.data_ptr() if noise_tape is not None else None,
                          const_noise=int(const_noise), philox_seed=philox_seed, sample_offset=sample_offset,
                          dump=dump.data_ptr() if dump is not None else None,
                          dump_steps=ds.ctypes.data if len(ds) else None, n_dump=len(ds), run_steps=run_steps,
                          k_base=k_base, clip_denoised=int(bool(clip_denoised)))
        _lib.check(self.lib.gdx_sample_loop(self.handle, C.byref(a), _stream(x.device)), self.lib)
        self._keep_loop = (tmap, ds)

    def forward_flops(self, mode=GDX_COND):
        f = C.c_double()
        _lib.check(self.lib.gdx_forward_flops(self.handle, mode, C.byref(f)), self.lib)
        return f.value

    def profile_begin(self, max_launches):
        _lib.check(self.lib.gdx_profile_begin(self.handle, max_launches), self.lib)

    def profile_end(self):
        us, n = C.c_float(), C.c_int32()
        _lib.check(self.lib.gdx_profile_end(self.handle, C.byref(us), C.byref(n)), self.lib)
        return us.value, n.value

    def bench_ffn_gemm(self, iters, device):
        us = C.c_float()
        _lib.check(self.lib.gdx_bench_ffn_gemm(self.handle, iters, C.byref(us), _stream(device)), self.lib)
        return us.value


# ---------------------------------------------------------------------- standalone kernels
def sampler_update(kind, coef, x, x0_cond, out, t=None, step_index=0, x0_uncond=None, scale=None, inpaint_mask=None,
                   inpaint_motion=None, noise=None, const_noise=False, philox_seed=0, sample_offset=0, rng_step=0,
                   pred_xstart=None, cond_grad=None, cond_coef=None, clip_denoised=False):
    lib = _lib.load()
    B, J, F, T = x.shape
    a = _lib.UpdateArgs(kind=kind, batch=B, njoints=J * F, frames=T, coef=coef.data_ptr(),
                        t=t.data_ptr() if t is not None else None, step_index=step_index, x=x.data_ptr(),
                        x0_cond=x0_cond.data_ptr(), x0_uncond=x0_uncond.data_ptr() if x0_uncond is not None else None,
                        scale=scale.data_ptr() if scale is not None else None,
                        inpaint_mask=inpaint_mask.data_ptr() if inpaint_mask is not None else None,
                        inpaint_motion=inpaint_motion.data_ptr() if inpaint_motion is not None else None,
                        noise=noise.data_ptr() if noise is not None else None, const_noise=int(const_noise),
                        philox_seed=philox_seed, sample_offset=sample_offset, rng_step=rng_step, out=out.data_ptr(),
                        pred_xstart=pred_xstart.data_ptr() if pred_xstart is not None else None,
                        cond_grad=cond_grad.data_ptr() if cond_grad is not None else None,
                        cond_coef=cond_coef.data_ptr() if cond_coef is not None else None,
                        clip_denoised=int(bool(clip_denoised)))
    _lib.check(lib.gdx_sampler_update(C.byref(a), _stream(x.device)), lib)
    return out


def plms_update(kind, coef, t, x, pred_xstart, eps=(), out=None):
    """gdx_plms_update: kind 0 eps, 6 Euler predictor, 1..4 Adams-Bashforth, 5 Euler corrector (include/gdx.h)."""
    lib = _lib.load()
    ref = pred_xstart
    if out is None:
        out = torch.empty_like(ref)
    a = _lib.PlmsArgs(kind=kind, batch=ref.shape[0], per_sample=ref.numel() // ref.shape[0], coef=coef.data_ptr(),
                      t=t.data_ptr(), step_index=0, x=x.data_ptr() if x is not None else None,
                      pred_xstart=pred_xstart.data_ptr(), out=out.data_ptr())
    for i, e in enumerate(eps):
        a.eps[i] = e.data_ptr()
    _lib.check(lib.gdx_plms_update(C.byref(a), _stream(ref.device)), lib)
    return out


def postprocess(sample, mean, std):
    """inv_transform + position / rotation split of a generated chunk on the device (reference
    sample/generate.py:132-146): sample [B, 6*nj, 1, T] -> (positions, rotations), each [B, nj, 3, T]."""
    lib = _lib.load()
    x = f32c(sample, "sample")
    B, J, F, T = x.shape
    if F != 1 or J % 6:
        raise ValueError(f"expected [B, 6*n_joints, 1, T], got {tuple(x.shape)}")
    mean = torch.as_tensor(mean, dtype=torch.float64, device=x.device).contiguous()
    std = torch.as_tensor(std, dtype=torch.float64, device=x.device).contiguous()
    if mean.numel() != J or std.numel() != J:
        raise ValueError("mean / std must have one entry per feature")
    pos = torch.empty(B, J // 6, 3, T, device=x.device, dtype=torch.float32)
    rot = torch.empty_like(pos)
    _lib.check(lib.gdx_postprocess(_ptr(x), _ptr(mean), _ptr(std), _ptr(pos), _ptr(rot), B, J // 6, T, _stream(x.device)), lib)
    return pos, rot


def q_sample(x_start, noise, coef, idx):
    lib = _lib.load()
    out = torch.empty_like(x_start)
    _lib.check(lib.gdx_q_sample(_ptr(x_start), _ptr(noise), _ptr(coef), idx, x_start.numel(), _ptr(out),
                                _stream(x_start.device)), lib)
    return out


def q_sample_t(x_start, noise, coef, t):
    """q_sample with per-sample timesteps (int64 [B] on the device)."""
    lib = _lib.load()
    out = torch.empty_like(x_start)
    B = x_start.shape[0]
    _lib.check(lib.gdx_q_sample_t(_ptr(x_start), _ptr(noise), _ptr(coef), _ptr(t), B, x_start.numel() // B, _ptr(out),
                                  _stream(x_start.device)), lib)
    return out


def masked_l2(a, b, mask):
    """Per-sample masked mean squared error (reference masked_l2); mask bool [B,1,1,T]."""
    lib = _lib.load()
    B, J, F, T = a.shape
    m = require_device(mask, "mask").to(torch.bool).reshape(B, T).contiguous()
    out = torch.empty(B, device=a.device, dtype=torch.float32)
    _lib.check(lib.gdx_masked_l2(_ptr(a), _ptr(b), _ptr(m), _ptr(out), B, J * F, T, _stream(a.device)), lib)
    return out


def randn(shape, device, philox_seed, sample_offset=0, rng_step=0):
    lib = _lib.load()
    out = torch.empty(shape, device=device, dtype=torch.float32)
    per = out.numel() // shape[0]
    _lib.check(lib.gdx_randn(_ptr(out), shape[0], per, philox_seed, sample_offset, rng_step, _stream(out.device)), lib)
    return out
