"""Stated parity tolerances of the compute modes, as fractions of max|reference output| (the reference samples in fp32).

SURVEY.md 8(d): fp32 path -- single forward 1e-4, end of a loop 1e-3 (measured 2e-6 / 4e-6); 16-bit MFMA modes (fp16, bf16) 2e-2
for forwards and whole loops WITHOUT guidance.  Under classifier-free guidance (`model/cfg_sampler.py:23-28` of the reference:
x0 = u + s (c - u) = (1 - s) u + s c) the prediction is a signed combination of two forwards whose rounding errors are
independent, so the bound a mode can state for it is the triangle inequality over the two terms:

    tol(dtype, s) = tol(dtype) * (|s| + |1 - s|)          (= tol(dtype) for 0 <= s <= 1; 4 x at the CLI's default s = 2.5)

The sampler update after it is linear in x0 with coefficients <= 1 and `clip_denoised` is 1-Lipschitz; whole loops are measured
not to amplify the per-step error.  Measured against the oracle (tools/fuzz_loops.py, profiles/r03*): fp32
<= 4e-6 and fp16 <= 3e-3 with or without guidance; bf16 see DESIGN.md 4b.  `include/gdx.h` (GDX_DTYPE_*) quotes these numbers."""

FORWARD_TOL = {"fp32": 1e-4, "fp16": 2e-2, "bf16": 2e-2}
LOOP_TOL = {"fp32": 1e-3, "fp16": 2e-2, "bf16": 2e-2}


def guidance_factor(scale):
    """|s| + |1 - s|: how much the guided blend (1 - s) u + s c can amplify independent errors of its two forwards."""
    if scale is None:
        return 1.0
    s = float(scale)
    return abs(s) + abs(1.0 - s)


def stated_tolerance(compute_dtype, guidance_scale=None, loop=True):
    """The tolerance the mode states for a forward (loop=False) or a whole sampling loop, relative to max|reference|.
    guidance_scale: the largest classifier-free guidance scale of the batch (None: no ClassifierFreeSampleModel)."""
    table = LOOP_TOL if loop else FORWARD_TOL
    if compute_dtype not in table:
        raise ValueError(f"unknown compute dtype {compute_dtype!r}")
    return table[compute_dtype] * guidance_factor(guidance_scale)
