"""Oracle: MDM denoiser forward, CPU fp32 restatement (test infrastructure only).

Functional re-statement, over a plain ``{name: tensor}`` parameter dict that uses
the reference's state-dict names, of

  * `model/mdm.py:105-224`      MDM.forward      ("V2": project_to_lat + RoPE + causal
                                                  local attention + token + RoPE + encoder)
  * `model/mdm_old.py:84-122`   MDM_Old.forward  ("V1": concat MFCC + input linear + token
                                                  + absolute sinusoidal PE + encoder)
  * `model/local_attention.py:43-62,92-172`  rotary tables / rotate_half / windowed attention
  * `model/cfg_sampler.py:23-28`  classifier-free guidance blend
  * torch 1.7.1 `nn.TransformerEncoderLayer` (post-norm, erf-GELU, eps 1e-5, no final
    norm) as instantiated at `model/mdm.py:90-96`; the arithmetic of that layer lives in
    torch (pinned pytorch=1.7.1, environment.yml:88), restated here from its documented
    algorithm: x = LN1(x + OutProj(softmax(QK^T/sqrt(hd)) V)); x = LN2(x + W2 gelu(W1 x)).

Every stage can be captured through ``taps`` (dict filled in place) so that the HIP
kernels can be checked stage by stage.  Pinned by tests/golden/forward_*.npz, which
were produced by the reference modules themselves (oracle/tools/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- tables
def sinusoidal_pe(d_model, max_len=5000):
    """reference model/mdm.py:277-289 (PositionalEncoding.__init__) -> [max_len, d]."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def rotary_freqs(n, dim):
    """reference model/local_attention.py:43-53 -> freqs [n, dim] (two copies of dim/2)."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(n).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return torch.cat((freqs, freqs), dim=-1)


def rotate_half(x):
    """reference model/local_attention.py:55-58: (x1, x2) halves -> (-x2, x1)."""
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rotary(x, freqs):
    """reference model/local_attention.py:60-62."""
    return (x * freqs.cos()) + (rotate_half(x) * freqs.sin())


# ----------------------------------------------------------------------------- blocks
def timestep_embed(p, timesteps, pe):
    """reference model/mdm.py:296-310 -> [B, d]."""
    h = F.linear(pe[timesteps], p["embed_timestep.time_embed.0.weight"], p["embed_timestep.time_embed.0.bias"])
    h = F.silu(h)
    return F.linear(h, p["embed_timestep.time_embed.2.weight"], p["embed_timestep.time_embed.2.bias"])


def seed_embed(p, seed, uncond):
    """reference model/mdm.py:125-127,242-250,382-392 (eval mode: mask only when uncond)."""
    bs = seed.shape[0]
    flat = seed.squeeze(2).reshape(bs, -1)
    if uncond:
        flat = torch.zeros_like(flat)
    return F.linear(flat, p["seed_pose_encoder.seed_embed.weight"], p["seed_pose_encoder.seed_embed.bias"])


def local_attention(x, window=10):
    """reference model/local_attention.py:92-172 with the ctor arguments of
    model/mdm.py:72-80 (causal, look_backward=1, look_forward=0, q=k=v=x, all-ones mask).

    x: [BH, n, e].  Query p attends keys max(0,(p//w-1)*w) .. p (own + previous window,
    causal); padded look-back keys of the first window are masked (:148-159).
    Written per window with dense masked softmax, like the reference.
    """
    bh, n, e = x.shape
    if n % window != 0:
        raise ValueError(f"sequence length {n} must be divisible by window {window}")  # einops raises in the reference
    scale = e ** -0.5
    nw = n // window
    out = torch.empty_like(x)
    for w in range(nw):
        q = x[:, w * window:(w + 1) * window]                       # [BH, 10, e]
        k0 = max(0, (w - 1) * window)
        k = x[:, k0:(w + 1) * window]                               # [BH, 10|20, e]
        sim = torch.einsum("bie,bje->bij", q, k) * scale
        qi = torch.arange(w * window, (w + 1) * window).view(-1, 1)
        kj = torch.arange(k0, (w + 1) * window).view(1, -1)
        sim = sim.masked_fill(qi < kj, -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1)
        out[:, w * window:(w + 1) * window] = torch.einsum("bij,bje->bie", attn, k)
    return out


def encoder_layer(p, prefix, x, nhead, taps=None):
    """One post-norm nn.TransformerEncoderLayer in eval mode.  x: [S, B, d]."""
    S, B, d = x.shape
    hd = d // nhead
    qkv = F.linear(x, p[prefix + "self_attn.in_proj_weight"], p[prefix + "self_attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    # [S, B, d] -> [B*H, S, hd]
    q = q.reshape(S, B * nhead, hd).transpose(0, 1)
    k = k.reshape(S, B * nhead, hd).transpose(0, 1)
    v = v.reshape(S, B * nhead, hd).transpose(0, 1)
    scores = torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(hd))
    attn = torch.softmax(scores, dim=-1)
    ctx = torch.bmm(attn, v)                                        # [B*H, S, hd]
    ctx = ctx.transpose(0, 1).reshape(S, B, d)
    sa = F.linear(ctx, p[prefix + "self_attn.out_proj.weight"], p[prefix + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + sa, (d,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], 1e-5)
    ff = F.linear(x, p[prefix + "linear1.weight"], p[prefix + "linear1.bias"])
    ff = F.gelu(ff)                                                 # exact erf form
    ff = F.linear(ff, p[prefix + "linear2.weight"], p[prefix + "linear2.bias"])
    x = F.layer_norm(x + ff, (d,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], 1e-5)
    if taps is not None:
        taps[prefix + "ctx"] = ctx
        taps[prefix + "out"] = x
    return x


def encoder(p, x, num_layers, nhead, taps=None):
    for l in range(num_layers):
        x = encoder_layer(p, f"seqTransEncoder.layers.{l}.", x, nhead, taps)
    return x


def _heads_split(x, bs, n, heads):
    """[n, B, d] -> [B*heads, n, d/heads]   (reference model/mdm.py:176-179)."""
    x = x.permute(1, 0, 2).reshape(bs, n, heads, -1).permute(0, 2, 1, 3)
    return x.reshape(bs * heads, n, -1)


def _heads_merge(x, bs, n, heads):
    """[B*heads, n, e] -> [n, B, d]   (reference model/mdm.py:190-192, 208-213)."""
    x = x.reshape(bs, heads, n, -1).permute(0, 2, 1, 3).reshape(bs, n, -1)
    return x.permute(1, 0, 2)


# ----------------------------------------------------------------------------- forwards
def mdm_forward(p, cfg, x, timesteps, y, taps=None):
    """V2: reference model/mdm.py:105-224.  x [B,J,1,T] -> [B,J,1,T]."""
    bs, njoints, nfeats, nframes = x.shape
    d, heads, cl = cfg["latent_dim"], cfg["num_heads"], cfg.get("cl_head", 8)
    pe = sinusoidal_pe(d)
    uncond = bool(y.get("uncond", False))
    if "seed" not in y:
        raise KeyError("seed")
    if "mfcc" not in y:
        raise NotImplementedError
    emb_seed = seed_embed(p, y["seed"], uncond)                     # [B, d]
    emb_t = timestep_embed(p, timesteps, pe)                        # [B, d]
    emb_audio = y["mfcc"].squeeze(2).permute(2, 0, 1)               # [T, B, 26]
    xin = x.permute(3, 0, 1, 2).reshape(nframes, bs, njoints * nfeats)
    emb_pose = F.linear(xin, p["input_process.poseEmbedding.weight"], p["input_process.poseEmbedding.bias"])
    coa = (emb_seed + emb_t).unsqueeze(0)                           # [1, B, d]
    embs = torch.cat((emb_pose, emb_audio, coa.repeat(nframes, 1, 1)), dim=2)
    xseq = F.linear(embs, p["project_to_lat.weight"], p["project_to_lat.bias"])   # [T, B, d]
    if taps is not None:
        taps["emb_pose"] = emb_pose
        taps["coa"] = coa
        taps["project_to_lat"] = xseq
    # rotary + causal local attention on cl heads of d/cl dims (mdm.py:176-194)
    xs = _heads_split(xseq, bs, nframes, cl)
    xs = apply_rotary(xs, rotary_freqs(nframes, d // cl))
    if taps is not None:
        taps["rope1"] = xs
    xs = local_attention(xs, window=10)
    xseq = _heads_merge(xs, bs, nframes, cl)
    if taps is not None:
        taps["local_attn"] = xseq
    # conditioning token + second rotary over T+1 positions (mdm.py:197-213)
    xseq = torch.cat((coa, xseq), dim=0)
    xs = _heads_split(xseq, bs, nframes + 1, cl)
    xs = apply_rotary(xs, rotary_freqs(nframes + 1, d // cl))
    xseq = _heads_merge(xs, bs, nframes + 1, cl)
    if taps is not None:
        taps["enc_in"] = xseq
    out = encoder(p, xseq, cfg["num_layers"], heads, taps)[1:]
    out = F.linear(out, p["output_process.poseFinal.weight"], p["output_process.poseFinal.bias"])
    return out.reshape(nframes, bs, njoints, nfeats).permute(1, 2, 3, 0)


def mdm_old_forward(p, cfg, x, timesteps, y, taps=None):
    """V1: reference model/mdm_old.py:84-122."""
    bs, njoints, nfeats, nframes = x.shape
    d, heads = cfg["latent_dim"], cfg["num_heads"]
    pe = sinusoidal_pe(d)
    uncond = bool(y.get("uncond", False))
    emb = timestep_embed(p, timesteps, pe) + seed_embed(p, y["seed"], uncond)     # [B, d]
    xc = torch.cat((x, y["mfcc"]), dim=1)                           # [B, J+26, 1, T]
    xin = xc.permute(3, 0, 1, 2).reshape(nframes, bs, -1)
    h = F.linear(xin, p["input_process.poseEmbedding.weight"], p["input_process.poseEmbedding.bias"])
    xseq = torch.cat((emb.unsqueeze(0), h), dim=0)                  # [T+1, B, d]
    xseq = xseq + pe[: nframes + 1].unsqueeze(1)
    if taps is not None:
        taps["enc_in"] = xseq
    out = encoder(p, xseq, cfg["num_layers"], heads, taps)[1:]
    out = F.linear(out, p["output_process.poseFinal.weight"], p["output_process.poseFinal.bias"])
    return out.reshape(nframes, bs, njoints, nfeats).permute(1, 2, 3, 0)


def forward(p, cfg, x, timesteps, y, taps=None):
    fn = mdm_forward if cfg.get("arch", "mdm") == "mdm" else mdm_old_forward
    return fn(p, cfg, x, timesteps, y, taps)


def cfg_forward(p, cfg, x, timesteps, y):
    """reference model/cfg_sampler.py:23-28."""
    y_u = dict(y)
    y_u["uncond"] = True
    out = forward(p, cfg, x, timesteps, y)
    out_u = forward(p, cfg, x, timesteps, y_u)
    return out_u + (y["scale"].view(-1, 1, 1, 1) * (out - out_u))
