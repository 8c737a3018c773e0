"""CPU-only tests of the host side: drop-in surface (state-dict keys, ctor kwargs, CLI flags,
factory), schedule / coefficient tables against the reference's golden tables, the C-ABI
library's exports, the no-fallback rule, and the world_size-2 shard + gather path on gloo."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, load_golden

TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
          "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
          "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
          "posterior_mean_coef1", "posterior_mean_coef2"]


def _model(arch, J, d, L=2, **over):
    from gesturediffusion_amd.model.mdm import MDM
    from gesturediffusion_amd.model.mdm_old import MDM_Old
    kw = dict(modeltype="", njoints=J, nfeats=1, translation=True, pose_rep="rot6d", glob=True, glob_rot=True,
              latent_dim=d, ff_size=1024, num_layers=L, num_heads=4, dropout=0.1, activation="gelu",
              data_rep="genea_vec", cond_mask_prob=0.1, clip_version="ViT-B/32", dataset="genea2023", use_text=False,
              mfcc_input=True, use_wav_enc=False, seed_poses=10, use_audio=False)
    kw.update(over)
    return (MDM if arch == "mdm" else MDM_Old)(**kw)


def _diffusion(resp, sched="cosine", var_small=True):
    from gesturediffusion_amd.diffusion import gaussian_diffusion as gd
    from gesturediffusion_amd.diffusion.respace import SpacedDiffusion, space_timesteps
    return SpacedDiffusion(use_timesteps=space_timesteps(1000, resp if resp else [1000]),
                           betas=gd.get_named_beta_schedule(sched, 1000), model_mean_type=gd.ModelMeanType.START_X,
                           model_var_type=gd.ModelVarType.FIXED_SMALL if var_small else gd.ModelVarType.FIXED_LARGE,
                           loss_type=gd.LossType.MSE)


# ------------------------------------------------------------------------------- drop-in surface
@pytest.mark.parametrize("arch,J,d", [("mdm", 263, 512), ("mdm_old", 263, 512), ("mdm", 498, 256)])
def test_state_dict_keys_and_shapes_match_reference(arch, J, d):
    want = {}
    for line in open(os.path.join(GOLDEN, "state_dict_keys.txt")):
        a, j, dd, key, shape = line.split()
        if (a, int(j), int(dd)) == (arch, J, d):
            want[key] = tuple(int(s) for s in shape.split("x"))
    got = {k: tuple(v.shape) for k, v in _model(arch, J, d).state_dict().items()}
    assert got == want


def test_load_model_wo_clip_roundtrip_and_unexpected_keys():
    from gesturediffusion_amd.utils.model_util import load_model_wo_clip
    m = _model("mdm", 32, 64)
    sd = {k: v.clone() for k, v in _model("mdm", 32, 64).state_dict().items()}
    load_model_wo_clip(m, sd)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])
    with pytest.raises(AssertionError):
        load_model_wo_clip(m, dict(sd, bogus=torch.zeros(1)))


def test_ctor_surface_and_attributes():
    from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
    m = _model("mdm", 32, 64, some_future_kwarg=1)          # unknown kwargs are swallowed by **kargs
    assert (m.njoints, m.nfeats, m.data_rep, m.cond_mask_prob, m.input_feats) == (32, 1, "genea_vec", 0.1, 32)
    assert m.rot2xyz(x=torch.ones(2), mask=None, pose_rep="xyz") is not None
    assert len(m.parameters_wo_clip()) == len(list(m.parameters()))
    assert m.eval() is m and m.train(False) is m
    w = ClassifierFreeSampleModel(m)
    assert (w.njoints, w.nfeats, w.data_rep) == (32, 1, "genea_vec") and w.rot2xyz is m.rot2xyz
    with pytest.raises(AssertionError):
        ClassifierFreeSampleModel(_model("mdm", 32, 64, cond_mask_prob=0.0))
    with pytest.raises(AttributeError):                      # reference: audio_feat_dim undefined (mdm.py:72)
        _model("mdm", 32, 64, mfcc_input=False)
    with pytest.raises(NotImplementedError):
        _model("mdm", 32, 64, use_text=True)


def test_no_cpu_fallback():
    from gesturediffusion_amd._lib import GdxError
    from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
    m = _model("mdm", 32, 64).eval()
    x = torch.zeros(2, 32, 1, 20)
    y = {"seed": torch.zeros(2, 32, 1, 10), "mfcc": torch.zeros(2, 26, 1, 20), "scale": torch.ones(2)}
    with pytest.raises(GdxError):
        m(x, torch.zeros(2, dtype=torch.long), y)
    with pytest.raises(GdxError):
        ClassifierFreeSampleModel(m)(x, torch.zeros(2, dtype=torch.long), y)
    with pytest.raises(GdxError):
        _diffusion([10]).p_sample_loop(m, (2, 32, 1, 20), clip_denoised=False, model_kwargs={"y": y})


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "gesturediffusion_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src, f


# ------------------------------------------------------------------------------- C ABI
def test_c_abi_exports_every_declared_symbol():
    from gesturediffusion_amd import _lib
    hdr = open(os.path.join(REPO, "include", "gdx.h")).read()
    declared = set(re.findall(r"\b(gdx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    assert declared == set(_lib.EXPORTS)
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libgdx.so not built (run python __graft_entry__.py)")
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True,
                          text=True).stdout
    for n in declared:
        assert f" T {n}\n" in syms, n
    lib = _lib.load()                                       # loads on a CPU-only box (no compute calls)
    assert lib.gdx_last_error() is not None


def test_c_abi_rejects_bad_config_without_gpu():
    import ctypes as C
    from gesturediffusion_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libgdx.so not built")
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.Config(arch=7, njoints=16, latent_dim=128, ff_size=256, num_layers=2, num_heads=4, seed_poses=10,
                      mfcc_dim=26, cl_head=8, window=10)
    assert lib.gdx_create(C.byref(cfg), C.byref(h)) != 0 and b"arch" in lib.gdx_last_error()
    cfg.arch, cfg.latent_dim = 2, 100
    assert lib.gdx_create(C.byref(cfg), C.byref(h)) != 0 and b"latent_dim" in lib.gdx_last_error()
    assert lib.gdx_sampler_update(None, None) != 0


# ------------------------------------------------------------------------------- schedule / coefficients
@pytest.mark.parametrize("sched", ["cosine", "linear"])
@pytest.mark.parametrize("tag,resp", [("1000", ""), ("ddim10", "ddim10"), ("ddim100", "ddim100"), ("s20", [20])])
def test_product_schedule_tables_bit_exact_vs_reference(sched, tag, resp):
    g = load_golden("schedule.npz")
    df = _diffusion(resp, sched)
    for n in TABLES:
        assert np.array_equal(getattr(df, n), g[f"{sched}.{tag}.{n}"]), n
    assert np.array_equal(np.array(df.timestep_map), g[f"{sched}.{tag}.timestep_map"])


def test_coef_tables_match_oracle_rounding():
    from gesturediffusion_amd._lib import GDX_SAMPLER_DDIM, GDX_SAMPLER_P
    from oracle import sampler as osamp
    from oracle import schedule as osch
    df = _diffusion("ddim100")
    tab, _ = osch.make_tables("cosine", 1000, "ddim100")
    t = torch.arange(tab.num_timesteps)
    c = df.coef_table(GDX_SAMPLER_P, "cpu")
    assert torch.equal(c[:, 0], osamp.extract(tab.posterior_mean_coef1, t).view(-1))
    assert torch.equal(c[:, 1], osamp.extract(tab.posterior_mean_coef2, t).view(-1))
    nz = (t != 0).float()
    assert torch.equal(c[:, 2], nz * torch.exp(0.5 * osamp.extract(tab.posterior_log_variance_clipped, t).view(-1)))
    cd = df.coef_table(GDX_SAMPLER_DDIM, "cpu", eta=0.5)
    ab, abp = osamp.extract(tab.alphas_cumprod, t).view(-1), osamp.extract(tab.alphas_cumprod_prev, t).view(-1)
    sigma = 0.5 * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    assert torch.equal(cd[:, 4], nz * sigma) and torch.equal(cd[:, 3], torch.sqrt(1 - abp - sigma ** 2))
    # FIXED_LARGE swaps the variance table (gaussian_diffusion.py:334-341)
    cl = _diffusion("ddim100", var_small=False).coef_table(GDX_SAMPLER_P, "cpu")
    assert not torch.equal(cl[:, 2], c[:, 2])


def test_unsupported_modes_raise():
    from gesturediffusion_amd.diffusion import gaussian_diffusion as gd
    df = gd.GaussianDiffusion(betas=gd.get_named_beta_schedule("linear", 1000), model_mean_type=gd.ModelMeanType.EPSILON,
                              model_var_type=gd.ModelVarType.LEARNED_RANGE, loss_type=gd.LossType.MSE)
    with pytest.raises(NotImplementedError):       # learned variances: no 2x-channel denoiser exists on the path
        df.coef_table(0, "cpu")
    with pytest.raises(NotImplementedError):       # the forward half is implemented for the configured mode only
        gd.GaussianDiffusion(betas=gd.get_named_beta_schedule("cosine", 1000), model_mean_type=gd.ModelMeanType.START_X,
                             model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE,
                             lambda_vel=1.0).training_losses(None, None, None, model_kwargs={"y": {"mask": None}})
    with pytest.raises(NotImplementedError):
        df.coef_table(1, "cpu")
    with pytest.raises(NotImplementedError):
        _diffusion([10]).ddim_sample_loop(None, (1, 1, 1, 1), dump_steps=[0])
    with pytest.raises(NotImplementedError):
        _diffusion([10]).ddim_sample_loop(None, (1, 1, 1, 1), const_noise=True)


# ------------------------------------------------------------------------------- CLI / factory
def test_generate_args_json_override(tmp_path):
    from gesturediffusion_amd.utils.model_util import create_model_and_diffusion
    from gesturediffusion_amd.utils.parser_util import generate_args
    ck = tmp_path / "run" / "model000100.pt"
    ck.parent.mkdir()
    ck.write_bytes(b"")
    json.dump({"dataset": "genea2023", "latent_dim": 64, "layers": 2, "cond_mask_prob": 0.1, "mfcc_input": True,
               "seed_poses": 10, "noise_schedule": "cosine", "sigma_small": True, "num_frames": 120},
              open(ck.parent / "args.json", "w"))
    a = generate_args(["--model_path", str(ck), "--latent_dim", "999", "--guidance_param", "3.0"])
    assert a.latent_dim == 64 and a.layers == 2 and a.dataset == "genea2023"     # overwritten from args.json
    assert a.guidance_param == 3.0 and a.seed == 10 and a.batch_size == 256      # user-side groups keep CLI/defaults
    model, diffusion = create_model_and_diffusion(a, None)
    assert model.njoints == 498 and model.latent_dim == 64 and model.ff_size == 1024 and model.num_heads == 4
    from gesturediffusion_amd.diffusion import gaussian_diffusion as gd
    assert diffusion.num_timesteps == 1000 and diffusion.model_mean_type == gd.ModelMeanType.START_X
    assert diffusion.model_var_type == gd.ModelVarType.FIXED_SMALL and diffusion.timestep_map == list(range(1000))
    json.dump({"dataset": "genea2023", "cond_mask_prob": 0.0}, open(ck.parent / "args.json", "w"))
    assert generate_args(["--model_path", str(ck)]).guidance_param == 1            # parser_util.py:31-32
    a = generate_args(["--synthetic", "--timestep_respacing", "ddim100", "--sigma_small", ""])
    assert a.sigma_small is False                                                  # type=bool quirk kept


def test_space_timesteps_matches_oracle():
    from gesturediffusion_amd.diffusion.respace import space_timesteps
    from oracle import schedule as osch
    for n, sc in [(1000, "ddim10"), (1000, "ddim100"), (1000, [10]), (300, "10,15,20"), (1000, [1000]), (1000, "ddim25")]:
        assert space_timesteps(n, sc) == osch.space_timesteps(n, sc)
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")


# ------------------------------------------------------------------------------- multi-GPU path on gloo
def test_shard_range_partitions():
    from gesturediffusion_amd.utils.dist_util import shard_range
    for total, world in [(2048, 8), (41, 8), (64, 1), (5, 8)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["GDX_REPO"])
from gesturediffusion_amd.utils import dist_util
from oracle import philox
rank, world, device = dist_util.init_from_env(backend="gloo")
J, T, SEED = 5, 6, 10
# each rank fills ITS shard with the noise the sampler would draw: keyed by the GLOBAL sample index, so the gathered
# batch must equal the batch one rank would have produced alone.  7 over 2 and 41 over 2 are uneven splits; 1 over 2
# leaves rank 1 with an empty shard.
for total in (7, 41, 8, 1):
    lo, hi = dist_util.shard_range(total, rank, world)
    local = torch.from_numpy(philox.normal(hi - lo, J * T, SEED, sample_offset=lo, step=3)).view(hi - lo, J, 1, T)
    full = dist_util.gather_samples(local, total)
    if rank == 0:
        want = torch.from_numpy(philox.normal(total, J * T, SEED, sample_offset=0, step=3)).view(total, J, 1, T)
        assert full.shape == want.shape and torch.equal(full, want), total
    else:
        assert full is None
try:
    dist_util.gather_samples(torch.zeros(3, J, 1, T), 41)          # wrong shard size is an error, not a hang
    raise SystemExit("expected ValueError")
except ValueError:
    pass
if rank == 0:
    print("GATHER_OK")
torch.distributed.barrier(); torch.distributed.destroy_process_group()
"""


def test_world_size_2_shard_and_gather_gloo(tmp_path):
    """Two ranks (gloo, CPU): shard_range + per-global-index fill + gather_samples == the unsharded batch."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, GDX_REPO=REPO, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, r.stdout + r.stderr


def test_gg_collate_matches_reference_fixture():
    """`data_loaders/tensors.py` of the reference, on ragged GENEA-style items (fixture made by oracle/tools/make_golden.py)."""
    sys.path.insert(0, os.path.join(REPO, "oracle", "tools"))
    from make_golden import collate_inputs
    from gesturediffusion_amd.data_loaders import tensors as ten
    z = np.load(os.path.join(GOLDEN, "collate.npz"))
    items = collate_inputs()
    nmin = min(x[2] for x in items)
    items = [(x[0], x[1], x[2], x[3][:nmin * 5], x[4][:nmin], x[5]) for x in items] + [None]
    motion, cond = ten.gg_collate([i for i in items if i is not None])
    y = cond["y"]
    assert set(y) == {"mask", "lengths", "text", "mfcc", "audio", "seed"}
    for name, got in (("motion", motion), ("mask", y["mask"]), ("lengths", y["lengths"]), ("mfcc", y["mfcc"]),
                      ("audio", y["audio"]), ("seed", y["seed"])):
        assert got.dtype == torch.from_numpy(z[name]).dtype, name
        assert np.array_equal(got.numpy(), z[name]), name
    assert y["text"] == list(z["text"])
    ragged = ten.collate_tensors([torch.ones(2, 3), torch.ones(1, 5) * 2, torch.ones(3, 1) * 3])
    assert np.array_equal(ragged.numpy(), z["ragged"])
    assert np.array_equal(ten.lengths_to_mask(torch.tensor([0, 3, 5]), 5).numpy(), z["len_mask"])
    # `collate` drops None items and falls back to the frame count when 'lengths' is missing
    m2, c2 = ten.collate([None, {"inp": torch.ones(3, 1, 4)}, {"inp": torch.ones(3, 1, 2)}])
    assert m2.shape == (2, 3, 1, 4) and c2["y"]["lengths"].tolist() == [4, 2]
    assert c2["y"]["mask"][1, 0, 0].tolist() == [True, True, False, False]


def test_packed_image_key_follows_checkpoint_content_arch_and_dtype(tmp_path):
    """utils/model_util.packed_image_path: the image of a checkpoint is keyed by the file's content, the model class and the
    compute dtype (no GPU needed for the key)."""
    from gesturediffusion_amd.utils.model_util import packed_image_path

    class MDM:
        compute_dtype = None

    class MDM_Old:
        compute_dtype = "fp16"
    a, b = tmp_path / "model000000001.pt", tmp_path / "copy.pt"
    a.write_bytes(b"weights-1" * 1000)
    b.write_bytes(b"weights-1" * 1000)
    k = packed_image_path("cache", str(a), MDM())
    assert k == packed_image_path("cache", str(b), MDM()) and k.endswith("-MDM-fp32.gdxpack")   # same content, other name
    assert packed_image_path("cache", str(a), MDM_Old()).endswith("-MDM_Old-fp16.gdxpack")
    a.write_bytes(b"weights-2" * 1000)
    assert packed_image_path("cache", str(a), MDM()) != k                                       # same name, other content


def test_bench_presets_are_the_baseline_configs():
    """bench.py --config N must be BASELINE.json's configuration N as SURVEY.md 8(d) spells it out (arch / J / B / T / d / L /
    respacing / guidance / dtype / how --gpus scales it), and the metric string must name the preset's own loop."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    want = {
        "1": dict(arch="mdm", J=150, d=512, L=8, T=60, batch=4, respacing="ddim10", sampler="ddim", cfg=False, dtype="fp32"),
        "2": dict(arch="mdm_old", J=263, d=512, L=8, T=196, batch=64, respacing="", sampler="p", cfg=False, dtype="fp32", global_batch=False),
        "3": dict(arch="mdm_old", J=263, d=512, L=8, T=196, batch=256, respacing="ddim100", sampler="ddim", cfg=True, dtype="fp32"),
        "4": dict(arch="mdm_old", J=263, d=512, L=8, T=196, batch=2048, sub=256, respacing="", sampler="p", cfg=False, global_batch=True,
                  scaling="strong"),
        "5": dict(arch="mdm", J=498, d=1024, L=8, T=520, batch=128, respacing="", sampler="p", cfg=False, dtype="fp16", global_batch=True,
                  scaling="strong"),
    }
    for name, fields in want.items():
        for k, v in fields.items():
            assert bench.PRESETS[name][k] == v, (name, k)
    assert bench.PRESETS["2"]["scaling"] == "weak"          # the headline: 64 samples per GPU
    assert bench.SCHEDULE_STEPS == 1000 and bench.F32_MFMA_PEAK_TFLOPS == 157.3 and bench.F16_MFMA_PEAK_TFLOPS == 2500.0



def test_bench_gpus_n_spawns_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` started plainly (no RANK in the environment) must start two ranks itself; on a box without
    a GPU both refuse to run (no CPU fallback) and the parent exits non-zero instead of printing a record."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert r.stderr.count("needs an MI355X") == 2 and "stopping the other" in r.stderr


def test_too_many_ranks_is_refused_on_every_rank():
    from gesturediffusion_amd.utils import dist_util
    dist_util.check_world(8, 8)
    with pytest.raises(ValueError, match="every rank needs at least one"):
        dist_util.check_world(5, 8)


def test_stated_tolerance_under_guidance():
    from gesturediffusion_amd.numerics import stated_tolerance
    assert stated_tolerance("fp16") == stated_tolerance("bf16", 1.0) == 2e-2
    assert stated_tolerance("bf16", 2.5) == pytest.approx(8e-2) and stated_tolerance("fp32", None, loop=False) == 1e-4
    with pytest.raises(ValueError):
        stated_tolerance("fp8")


def test_noise_block_is_sized_by_bytes():
    """The torch-generator seam pre-draws at most 256 MiB of noise per block (config 2: 50 x 13.2 MB would be 660 MB)."""
    from gesturediffusion_amd.diffusion.gaussian_diffusion import NOISE_BLOCK, NOISE_BLOCK_BYTES, noise_block_steps
    assert noise_block_steps(1000, 2 * 16 * 20) == NOISE_BLOCK                      # tiny tensors: the step cap
    c2 = noise_block_steps(1000, 64 * 263 * 196)
    assert c2 == NOISE_BLOCK_BYTES // (4 * 64 * 263 * 196) == 20 and c2 * 4 * 64 * 263 * 196 <= NOISE_BLOCK_BYTES
    assert noise_block_steps(1000, 128 * 498 * 520) == 2                            # config 5
    assert noise_block_steps(1000, 10 ** 10) == 1 and noise_block_steps(7, 10) == 7
