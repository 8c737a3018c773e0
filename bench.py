#!/usr/bin/env python3
"""Headline benchmark: denoised motion frames/sec of the 1000-step p_sample_loop
(BASELINE.json metric; config 2: HumanML3D-shape MDM enc-512, 263x1x196, batch 64, d=512).

  python bench.py [--gpus N --steps K --warmup W]          one JSON line on rank 0

A "step" is one denoising step of the sampling loop (one pass of the hot path -- denoiser
forward + fused sampler update -- over the whole batch).  The default K=1000 times exactly ONE
complete 1000-step p_sample_loop, so `value` = B*T / wall of a full loop with nothing
extrapolated; for other K the last K steps of the 1000-step schedule are timed and `value` is
normalised to a 1000-step loop (B*T / (1000 * seconds per step)), which is stated in `config`.
Inputs (x_T from in-kernel Philox, seed poses, MFCCs) and weights are resident in HBM before the
timed region.  N > 1: one process per GPU (torch.distributed.run), each rank samples its own 64
independent samples (weak scaling, no data-path collective), one RCCL gather of the finished
samples at the end inside the timed region.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LOOP_STEPS = 1000
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
F16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA (v_mfma_f32_16x16x32_f16)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Host cores this process may actually use (affinity and cgroup quota, not the machine total)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def build_model(arch, J, d, layers, device, seed=0):
    from gesturediffusion_amd.model.mdm import MDM
    from gesturediffusion_amd.model.mdm_old import MDM_Old
    from gesturediffusion_amd.utils.init import init_state_dict
    cfg = dict(arch=arch, njoints=J, nfeats=1, latent_dim=d, ff_size=1024, num_layers=layers, num_heads=4,
               seed_poses=10)
    kw = dict(njoints=J, nfeats=1, translation=True, pose_rep="rot6d", glob=True, glob_rot=True, latent_dim=d,
              ff_size=1024, num_layers=layers, num_heads=4, dropout=0.1, activation="gelu", data_rep="genea_vec",
              cond_mask_prob=0.1, dataset="genea2023", use_text=False, mfcc_input=True, use_wav_enc=False,
              seed_poses=10, use_audio=False)
    m = (MDM if arch == "mdm" else MDM_Old)(**kw)
    sd = init_state_dict(cfg, seed=seed)
    m.load_state_dict(sd, strict=False)
    m.to(device).eval()
    return m, cfg, sd


def make_diffusion():
    from gesturediffusion_amd.diffusion import gaussian_diffusion as gd
    from gesturediffusion_amd.diffusion.respace import SpacedDiffusion, space_timesteps
    return SpacedDiffusion(use_timesteps=space_timesteps(LOOP_STEPS, [LOOP_STEPS]),
                           betas=gd.get_named_beta_schedule("cosine", LOOP_STEPS),
                           model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                           loss_type=gd.LossType.MSE)


def cpu_baseline(cfg, sd, B, T, seedp, mfcc, steps=6):
    """The reference's CPU path, restated (oracle: same torch-CPU ops the reference dispatches),
    on this box's host cores: `steps` denoise steps of the SAME workload (full batch), extrapolated
    linearly to the 1000-step loop."""
    from oracle import mdm_forward as omf
    from oracle import sampler as osamp
    from oracle import schedule as osch
    cores = usable_cores()
    torch.set_num_threads(cores)
    tab, tmap = osch.make_tables("cosine", LOOP_STEPS, "")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, cfg["njoints"], 1, T, generator=g)
    y = {"seed": seedp, "mfcc": mfcc}
    mapt = torch.tensor(tmap)

    def one(i, x):
        t = torch.tensor([i] * B)
        with torch.no_grad():
            x0 = omf.forward(sd, cfg, x, mapt[t], y)
            return osamp.p_sample_step(tab, x0, x, t, torch.randn(x.shape, generator=g))
    log(f"cpu_baseline: {cores} threads, warm-up step ...")
    x = one(LOOP_STEPS - 1, x)       # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    for k in range(steps):
        x = one(LOOP_STEPS - 2 - k, x)
        log(f"cpu_baseline: step {k + 1}/{steps} ({time.perf_counter() - t0:.1f} s)")
    sec = (time.perf_counter() - t0) / steps
    return {"value": B * T / (LOOP_STEPS * sec), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{steps} of {LOOP_STEPS} denoise steps at the full batch ({B}x{cfg['njoints']}x1x{T}), "
                      f"{sec * 1e3:.0f} ms/step on torch-CPU, extrapolated linearly",
            "ms_per_step": sec * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=LOOP_STEPS)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--arch", default="mdm_old", choices=["mdm", "mdm_old"])
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--frames", type=int, default=0, help="default 196 (mdm_old) / 200 (mdm: T %% 10 == 0)")
    ap.add_argument("--njoints", type=int, default=263)
    ap.add_argument("--latent_dim", type=int, default=512)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--cfg", action="store_true", help="ClassifierFreeSampleModel (cond+uncond double batch)")
    ap.add_argument("--sampler", default="p", choices=["p", "ddim"])
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16"],
                    help="fp32: exact fp32 MFMA (BASELINE's headline path); fp16: fp16 MFMA operands, fp32 accumulate (config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=24)
    args = ap.parse_args()

    from gesturediffusion_amd.utils import dist_util
    from gesturediffusion_amd.utils.init import synthetic_inputs
    rank, world, device = dist_util.init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: the native path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    T = args.frames or (196 if args.arch == "mdm_old" else 200)
    B, J = args.batch, args.njoints

    model, cfg, sd = build_model(args.arch, J, args.latent_dim, args.layers, device)
    model.compute_dtype = args.dtype
    inner = model
    if args.cfg:
        from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
        model = ClassifierFreeSampleModel(model)
    df = make_diffusion()
    # every rank conditions on its own shard of one global synthetic batch
    _, seed_all, mfcc_all = synthetic_inputs(cfg, B * world, T, seed=10)
    lo = rank * B
    seedp, mfcc = seed_all[lo:lo + B], mfcc_all[lo:lo + B]
    y = {"seed": seedp.to(device), "mfcc": mfcc.to(device)}
    if args.cfg:
        y["scale"] = torch.full((B,), 2.5, device=device)
    fn = df.p_sample_loop if args.sampler == "p" else df.ddim_sample_loop

    def run(nsteps):
        """The last `nsteps` steps of the 1000-step schedule (all of it when nsteps == 1000)."""
        out = None
        left = nsteps
        while left > 0:
            n = min(left, LOOP_STEPS)
            out = fn(model, (B, J, 1, T), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=LOOP_STEPS - n,
                     rng="philox", philox_seed=10, sample_offset=lo)
            left -= n
        full = dist_util.gather_samples(out, B * world)     # end-of-loop gather (RCCL over xGMI when N > 1)
        return full if full is not None else out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model on {device}, warm-up {args.warmup} steps ...")
    if args.warmup > 0:
        run(args.warmup)
    barrier()
    log(f"timing {args.steps} steps ...")
    eng0 = inner._get_engine(device)
    if rank == 0:
        eng0.profile_begin(512)      # HIP events around the first 512 FFN-1 GEMM launches of the timed region
    t0 = time.perf_counter()
    out = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all()

    ms_per_step = elapsed * 1e3 / args.steps
    log(f"{elapsed:.3f} s for {args.steps} steps = {ms_per_step:.3f} ms/step")
    frames_per_sec = world * B * T / (LOOP_STEPS * ms_per_step * 1e-3)

    if rank == 0:
        from gesturediffusion_amd.engine import GDX_CFG, GDX_COND
        eng = inner._get_engine(device)
        flops_step = eng.forward_flops(GDX_CFG if args.cfg else GDX_COND)
        # dominant kernel = the fp32 MFMA GEMM; its north-star instance is FFN linear1 (+bias+GELU):
        # algorithmic FLOPs per launch = 2 * (B*(T+1)) * d * ff, timed with HIP events on the launch stream
        N, d, ff = (2 if args.cfg else 1) * B * (T + 1), args.latent_dim, 1024
        gemm_us, gemm_launches = eng.profile_end()
        gemm_flops = 2.0 * N * d * ff
        traffic, traffic_src = None, None
        f16_mode = args.dtype == "fp16"
        pmc = os.path.join(REPO, "profiles", "r01g_pmc_ffn1_traffic.json")
        if os.path.exists(pmc) and not args.cfg and not f16_mode and (B, T, d, args.arch) == (64, 196, 512, "mdm_old"):
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]     # PMC passes cannot run inside this process
            traffic_src = "profiles/r01g_pmc_ffn1_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench)"
        achieved = gemm_flops / (gemm_us * 1e-6) / 1e12
        f16 = args.dtype == "fp16"
        peak = F16_MFMA_PEAK_TFLOPS if f16 else F32_MFMA_PEAK_TFLOPS
        rec = {
            "metric": "denoised motion frames/sec (1000-step p_sample_loop, B=64, T=196, d=512)",
            "value": round(frames_per_sec, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (fp32 accumulate)" if f16 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {5 if args.latent_dim == 1024 else 2}: {'MDM_Old (V1 encoder-only topology)' if args.arch == 'mdm_old' else 'MDM (V2)'}"
                            f" J={J} d={args.latent_dim} ff=1024 L={args.layers} H=4, "
                            f"{LOOP_STEPS}-step {'p_sample_loop' if args.sampler == 'p' else 'ddim_sample_loop'}"
                            f"{' + CFG' if args.cfg else ''}, batch {B}/GPU x {T} frames, random weights, Philox noise",
                "global_batch": B * world, "frames": T, "parallelism": f"dp{world} (independent samples, end-of-loop gather)",
                "step": "one denoising step (forward + fused sampler update) over the batch",
                "timed": ("one complete 1000-step loop" if args.steps == LOOP_STEPS else
                          f"last {args.steps} steps of the 1000-step schedule; value normalised to a 1000-step loop"),
            },
            "loop_seconds": round(ms_per_step * LOOP_STEPS * 1e-3, 3),
            "frame_steps_per_sec": round(world * B * T / (ms_per_step * 1e-3), 1),
            "step_tflops": round(world * flops_step / (ms_per_step * 1e-3) / 1e12, 2),
            ("step_frac_of_f16_mfma_peak" if f16 else "step_frac_of_f32_mfma_peak"):
                round(flops_step / (ms_per_step * 1e-3) / 1e12 / peak, 4),
            "roofline": {"bound": "mfma", "kernel": ("gemmh_kernel (gemmh.hip)" if f16 else "gemm4_kernel (gemm2.hip)") +
                                                    ", FFN linear1 + bias + GELU, "
                                                    f"M={N} N={ff} K={d}; HIP events on the launch stream around "
                                                    f"{gemm_launches} launches inside the timed loop",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "avg_launch_us": round(gemm_us, 2), "flops_per_launch": gemm_flops},
        }
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(cfg, sd, B, T, seedp, mfcc, args.cpu_steps)
            rec["gpu_over_cpu"] = round(frames_per_sec / world / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
