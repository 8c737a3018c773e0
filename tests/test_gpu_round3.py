"""Round-3 parity tests (need an MI355X): the stated tolerance of the 16-bit modes under classifier-free guidance, with the
guided + clipped bf16 loops that left the 2e-2 band in round 2 INSIDE the test; bench.py starting its own ranks."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO, rel_err
from test_gpu_parity import TINY, _diffusion, build_model, dev

pytestmark = pytest.mark.gpu


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_loops", os.path.join(REPO, "tools", "fuzz_loops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ bf16 under guidance
# Every guided + clip_denoised bf16 case of `tools/fuzz_loops.py 150 2` and `150 3` -- the two sweeps of
# profiles/r02n_fuzz_final_build.txt in which six such loops ended at 2.1e-2 .. 2.9e-2 of max|ref| (the band of the mode is
# 2e-2 without guidance).  No selection: all fifteen run.
ROUND2_SWEEPS = [(150, 2), (150, 3)]
BF16_GUIDED_REGRESSION_GUARD = 2e-2       # the mode's plain band: with the fp32 residual stream (round 3) these loops measure
                                          # <= 1.6e-2 (round 2, 16-bit stream: up to 2.9e-2) -- profiles/r03a_bf16_stream32_ab.txt


def test_bf16_guided_clipped_loops_of_the_round2_sweeps_meet_the_stated_bound():
    """Guided + clipped bf16 loops against the oracle's loop on the same noise tape.  The mode states
    tol(s) = 2e-2 * (|s| + |1 - s|) of max|ref| under a guidance scale s (gesturediffusion_amd/numerics.py: the guided
    prediction (1 - s) u + s c combines two forwards with independent rounding errors): asserted per case with the batch's
    largest scale, and -- since round 3 keeps bf16's residual stream in fp32 -- the plain 2e-2 of the mode as well."""
    from gesturediffusion_amd.numerics import stated_tolerance
    fz = _fuzz()
    ran, worst = 0, 0.0
    for n, seed in ROUND2_SWEEPS:
        for c in fz.draw_cases(n, seed):
            if not (c["dtype"] == "bf16" and c["cfg"] and c["clip"]):
                continue
            err = fz.run_case(c)
            ran += 1
            worst = max(worst, err)
            assert err < stated_tolerance("bf16", c["scale_max"]), (seed, fz.describe(c), err)
            assert err < BF16_GUIDED_REGRESSION_GUARD, (seed, fz.describe(c), err)
    assert ran == 15
    print(f"bf16 guided + clipped loops of the round-2 sweeps: worst {worst:.2e}")


@pytest.mark.parametrize("arch", ["mdm", "mdm_old"])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("kind", ["p", "ddim"])
def test_guided_clipped_loops_at_scale_2p5_meet_the_stated_bound(arch, dtype, kind):
    """Both topologies, every sample at the CLI's default guidance 2.5 (`utils/parser_util.py:153`), clip_denoised=True,
    ancestral and DDIM (eta 0.5), 20 steps: <= stated_tolerance(dtype, 2.5) of max|ref| against the oracle."""
    from gesturediffusion_amd.numerics import stated_tolerance
    fz = _fuzz()
    T = 40 if arch == "mdm" else 33
    c = dict(case=900 + (arch == "mdm") + 2 * (kind == "p"), arch=arch, J=37, T=T, B=3, steps=20, kind=kind,
             eta=0.5 if kind == "ddim" else 0.0, cfg=True, clip=True, const=False, skip=0, init=False, inpaint=False, dump=None,
             dtype=dtype, scale_max=2.5)
    err = fz.run_case(c, scale=2.5)
    assert err < stated_tolerance(dtype, 2.5), err
    if dtype == "bf16":
        assert err < BF16_GUIDED_REGRESSION_GUARD, err
    else:
        assert err < 1e-2, err


def test_unguided_and_low_scale_bf16_loops_keep_the_plain_2e2():
    """Without guidance, and for scales in [0, 1] (a convex combination of the two forwards), the bound is the mode's 2e-2."""
    from gesturediffusion_amd.numerics import stated_tolerance
    assert stated_tolerance("bf16", None) == stated_tolerance("bf16", 0.7) == 2e-2
    assert abs(stated_tolerance("bf16", 2.5) - 8e-2) < 1e-12
    fz = _fuzz()
    for arch, T in (("mdm", 30), ("mdm_old", 36)):
        base = dict(case=950, arch=arch, J=18, T=T, B=2, steps=12, kind="p", eta=0.0, clip=True, const=False, skip=0,
                    init=False, inpaint=False, dump=None, dtype="bf16")
        assert fz.run_case(dict(base, cfg=False, scale_max=0.0)) < 2e-2
        assert fz.run_case(dict(base, cfg=True, scale_max=0.7), scale=0.7) < 2e-2


# ------------------------------------------------------------------------------------------------ bench.py starts its own ranks
SMALL = ["--config", "4", "--batch", "6", "--latent_dim", "128", "--layers", "2", "--frames", "20", "--njoints", "37",
         "--respacing", "25", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]


def _bench(extra, env=None, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + SMALL + extra, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_starts_its_own_ranks_and_gathers_the_one_rank_samples(tmp_path):
    """`python bench.py --gpus 2` with NO launcher (plain python, no RANK in the environment): the parent spawns the two
    ranks itself (here both on this one GPU, gloo for the gather), exits 0, rank 0 prints ONE JSON line with n_gpus 2 and the
    gathered samples are bit-equal to the one-rank run of the same global batch (Philox keyed by the global sample index)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    one = _bench(["--gpus", "1", "--save-samples", str(tmp_path / "one.pt")], env=env)
    two = _bench(["--gpus", "2", "--save-samples", str(tmp_path / "two.pt")],
                 env=dict(env, GDX_SINGLE_GPU_RANKS="1", GDX_DIST_BACKEND="gloo"))
    assert one["n_gpus"] == 1 and one["dist"]["world"] == 1 and one["dist"]["backend"] is None
    assert two["n_gpus"] == 2 and two["dist"]["world"] == 2 and two["dist"]["backend"] == "gloo"
    assert two["dist"]["launcher"].startswith("bench.py") and [r["rank"] for r in two["dist"]["ranks"]] == [0, 1]
    assert two["scaling"] == "strong" and two["config"]["global_batch"] == one["config"]["global_batch"] == 6
    a, b = torch.load(tmp_path / "one.pt"), torch.load(tmp_path / "two.pt")
    assert a.shape == b.shape == (6, 37, 1, 20) and torch.equal(a, b)


def test_bench_refuses_more_ranks_than_samples():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    args = [a if a != "6" else "1" for a in SMALL]
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args + ["--gpus", "2"],
                       env=dict(env, GDX_SINGLE_GPU_RANKS="1", GDX_DIST_BACKEND="gloo"), capture_output=True, text=True,
                       timeout=600, cwd=REPO)
    assert r.returncode != 0 and "every rank needs at least one" in r.stderr
