#!/usr/bin/env python3
"""Benchmark of the sampling hot path: denoised motion frames/sec of one complete sampling loop.

  python bench.py [--gpus N --steps K --warmup W] [--config {1,2,3,4,5,genea}]      one JSON line on rank 0

The default (no --config) is BASELINE.json's headline: config 2 -- HumanML3D-shape MDM enc-512 (263x1x196), 1000-step
p_sample_loop, batch 64 per GPU, fp32 -- and with --gpus N it is weak scaling (64 samples per GPU).  `--config` selects
one of BASELINE.json's other configurations exactly as SURVEY.md 8(d) lists them:

  1      V2, J=150, B=4, T=60, d=512, L=8, 10-step DDIM (eta 0); the CPU leg is timed IN FULL (whole loops)
  2      V1 topology, J=263, B=64/GPU, T=196, d=512, L=8, 1000-step ancestral loop                 (weak scaling)
  3      config 2 + ClassifierFreeSampleModel (512 model rows per step), 100-step DDIM, B=256/GPU     (weak scaling)
  4      config 2 with a GLOBAL batch of 2048 split over --gpus (2048/N per rank, run as sub-batches of 256)  (strong)
  5      V2, J=498, d=1024, T=520, fp16 MFMA mode, GLOBAL batch 128 split over --gpus (16 per rank at 8)      (strong)
  genea  the CLI's real workload: V2, J=498, d=256, B=41, T=120, guidance 2.5, 1000-step ancestral loop (one chunk)

A "step" is one denoising step (one pass of the hot path -- denoiser forward [x2 rows under guidance] + fused sampler
update -- over the rank's batch).  K defaults to the loop length, so the default run times exactly ONE complete loop with
nothing extrapolated; for other K the last K steps of the schedule are timed and `value` is normalised to one loop
(B*T / (loop_steps * seconds per step)), which `config.timed` states.  Inputs (x_T, seed poses, MFCCs) and weights are
resident in HBM before the timed region.  N > 1: one process per GPU, no data-path collective, ONE gather of the finished
samples at the end, inside the timed region.  The ranks come either from a launcher (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when
`python bench.py --gpus N` is started plainly (no RANK in the environment), from bench.py itself: the parent starts N fresh
children with that environment, never touches the GPU, and exits non-zero if any child did.  The record's `dist` object says
which backend / world size / devices the ranks saw.

`--seam` selects which side of the drop-in boundary drives the loop: `philox` (default; in-kernel counter-based noise),
`torch` (the reference caller's own kwargs -- torch's generator, progress=True -- which also run inside libgdx) or
`stepwise` (the model(x, t, **kw) callable protocol: one Python call + one update launch per step).
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

SCHEDULE_STEPS = 1000
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
F16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA
TRAFFIC_FILE = os.path.join(REPO, "profiles", "pmc_ffn1_traffic.json")   # {config: {"hbm_bytes_per_launch": ..., "source": ...}}

# arch, J, d, L, T, batch, batch_is_global, sub_batch, respacing, sampler, guidance, dtype, scaling
PRESETS = {
    "1": dict(arch="mdm", J=150, d=512, L=8, T=60, batch=4, global_batch=False, sub=0, respacing="ddim10", sampler="ddim",
              cfg=False, dtype="fp32", scaling="weak", label="BASELINE config 1"),
    "2": dict(arch="mdm_old", J=263, d=512, L=8, T=196, batch=64, global_batch=False, sub=0, respacing="", sampler="p",
              cfg=False, dtype="fp32", scaling="weak", label="BASELINE config 2"),
    "3": dict(arch="mdm_old", J=263, d=512, L=8, T=196, batch=256, global_batch=False, sub=0, respacing="ddim100",
              sampler="ddim", cfg=True, dtype="fp32", scaling="weak", label="BASELINE config 3"),
    "4": dict(arch="mdm_old", J=263, d=512, L=8, T=196, batch=2048, global_batch=True, sub=256, respacing="", sampler="p",
              cfg=False, dtype="fp32", scaling="strong", label="BASELINE config 4"),
    "5": dict(arch="mdm", J=498, d=1024, L=8, T=520, batch=128, global_batch=True, sub=0, respacing="", sampler="p",
              cfg=False, dtype="fp16", scaling="strong", label="BASELINE config 5"),
    "genea": dict(arch="mdm", J=498, d=256, L=8, T=120, batch=41, global_batch=False, sub=0, respacing="", sampler="p",
                  cfg=True, dtype="fp32", scaling="weak", label="GENEA chunk (sample.generate's workload)"),
}


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


def make_diffusion(respacing):
    from gesturediffusion_amd.diffusion import gaussian_diffusion as gd
    from gesturediffusion_amd.diffusion.respace import SpacedDiffusion, space_timesteps
    return SpacedDiffusion(use_timesteps=space_timesteps(SCHEDULE_STEPS, respacing or [SCHEDULE_STEPS]),
                           betas=gd.get_named_beta_schedule("cosine", SCHEDULE_STEPS),
                           model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                           loss_type=gd.LossType.MSE)


def cpu_baseline(p, cfg, sd, B, seedp, mfcc, loop_steps, budget_s, full_loops):
    """The reference's CPU path, restated (oracle: the same torch-CPU ops the reference dispatches), on this box's host
    cores, on the SAME workload at the rank's full batch.  full_loops (config 1): whole loops are timed, nothing is
    extrapolated.  Otherwise a bounded number of denoise steps (about `budget_s` seconds of CPU work, at least 2) is
    timed and extrapolated linearly to the loop -- flagged in `sample`."""
    from oracle import mdm_forward as omf
    from oracle import sampler as osamp
    from oracle import schedule as osch
    cores = usable_cores()
    torch.set_num_threads(cores)
    T, J = p["T"], cfg["njoints"]
    tab, tmap = osch.make_tables("cosine", SCHEDULE_STEPS, p["respacing"])
    g = torch.Generator().manual_seed(0)
    y = {"seed": seedp, "mfcc": mfcc}
    if p["cfg"]:
        y["scale"] = torch.full((B,), 2.5)
    fwd = (lambda x, t, yy: omf.cfg_forward(sd, cfg, x, t, yy)) if p["cfg"] else (lambda x, t, yy: omf.forward(sd, cfg, x, t, yy))
    mapt = torch.tensor(tmap)

    def one(i, x):
        t = torch.tensor([i] * B)
        with torch.no_grad():
            x0 = fwd(x, mapt[t], y)
            z = torch.randn(x.shape, generator=g)
            return osamp.p_sample_step(tab, x0, x, t, z) if p["sampler"] == "p" else osamp.ddim_step(tab, x0, x, t, z, 0.0)
    log(f"cpu_baseline: {cores} threads, warm-up step ...")
    x = torch.randn(B, J, 1, T, generator=g)
    t0 = time.perf_counter()
    x = one(loop_steps - 1, x)       # warm-up (thread pool, allocator); also sizes the sample
    first = time.perf_counter() - t0
    if full_loops:
        loops = max(1, min(20, int(budget_s / max(first * loop_steps, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(loops):
            x = torch.randn(B, J, 1, T, generator=g)
            for i in range(loop_steps - 1, -1, -1):
                x = one(i, x)
        sec = (time.perf_counter() - t0) / (loops * loop_steps)
        sample = f"{loops} complete {loop_steps}-step loop(s) at the full batch ({B}x{J}x1x{T}), timed in full: {sec * 1e3:.1f} ms/step on torch-CPU"
    else:
        steps = max(2, min(loop_steps - 1, int(budget_s / max(first, 1e-3))))
        t0 = time.perf_counter()
        for k in range(steps):
            x = one(loop_steps - 2 - k, x)
            log(f"cpu_baseline: step {k + 1}/{steps} ({time.perf_counter() - t0:.1f} s)")
        sec = (time.perf_counter() - t0) / steps
        sample = (f"{steps} of {loop_steps} denoise steps at the full batch ({B}x{J}x1x{T}"
                  f"{', guidance: 2 model passes per step' if p['cfg'] else ''}), {sec * 1e3:.0f} ms/step on torch-CPU, "
                  f"extrapolated linearly")
    return {"value": B * T / (loop_steps * sec), "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample,
            "ms_per_step": sec * 1e3}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh children (one per GPU) with the rendezvous environment of
    torch.distributed.run and wait for them.  The parent never initialises the GPU and never re-execs; a child that fails
    takes the others down (exact PIDs) so that nobody is left waiting in a barrier."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GDX_BENCH_LAUNCHER="bench.py (self-spawned ranks)")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL between processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    while procs:
        time.sleep(0.2)
        for pr in list(procs):
            code = pr.poll()
            if code is None:
                continue
            procs.remove(pr)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                log(f"a rank exited with code {code}: stopping the other {len(procs)}")
                for other in procs:
                    other.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed denoising steps (default: one complete loop)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="2", choices=sorted(PRESETS), help="BASELINE.json configuration (default 2 = headline)")
    # overrides of the preset (measurement sessions); the record then says 'custom'
    ap.add_argument("--arch", default=None, choices=["mdm", "mdm_old"])
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (global for configs 4 / 5)")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--njoints", type=int, default=0)
    ap.add_argument("--latent_dim", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--cfg", action="store_true", help="ClassifierFreeSampleModel (cond+uncond double batch)")
    ap.add_argument("--sampler", default=None, choices=["p", "ddim"])
    ap.add_argument("--respacing", default=None)
    ap.add_argument("--dtype", default=None, choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--seam", default="philox", choices=["philox", "torch", "stepwise"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work the cpu_baseline leg may spend")
    ap.add_argument("--save-samples", default=None, help="rank 0 writes the gathered samples of the timed run here (torch.save)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))       # before anything touches the GPU
    # stdout carries exactly ONE line, rank 0's record: whatever libraries print there (gloo's connection banner, RCCL
    # notices) goes to stderr instead
    sys.stdout.flush()
    record_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    p = dict(PRESETS[args.config])
    custom = False
    for key, val in (("arch", args.arch), ("batch", args.batch), ("T", args.frames), ("J", args.njoints), ("d", args.latent_dim),
                     ("L", args.layers), ("sampler", args.sampler), ("respacing", args.respacing), ("dtype", args.dtype)):
        if val:
            custom = custom or p[key] != val
            p[key] = val
    if args.cfg and not p["cfg"]:
        p["cfg"], custom = True, True

    from gesturediffusion_amd.utils import dist_util
    from gesturediffusion_amd.utils.init import synthetic_inputs
    if p["global_batch"]:
        try:
            dist_util.check_world(p["batch"])
        except ValueError as e:
            raise SystemExit(str(e))
    rank, world, device = dist_util.init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: the native path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    T, J = p["T"], p["J"]
    if p["global_batch"]:
        total = p["batch"]
        lo, hi = dist_util.shard_range(total, rank, world)
    else:
        total = p["batch"] * world
        lo, hi = rank * p["batch"], (rank + 1) * p["batch"]
    B = hi - lo                                            # this rank's samples
    sub = min(B, p["sub"]) if p["sub"] else B              # samples per loop invocation (config 4: 256)

    model, cfg, sd = build_model(p["arch"], J, p["d"], p["L"], device)
    model.compute_dtype = p["dtype"]
    inner = model
    if p["cfg"]:
        from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
        model = ClassifierFreeSampleModel(model)
    df = make_diffusion(p["respacing"])
    loop_steps = df.num_timesteps
    steps = args.steps or loop_steps
    # every rank conditions on its own shard of one global synthetic batch
    _, seed_all, mfcc_all = synthetic_inputs(cfg, total, T, seed=10)
    seedp, mfcc = seed_all[lo:hi], mfcc_all[lo:hi]
    seed_d, mfcc_d = seedp.to(device), mfcc.to(device)
    scale_d = torch.full((B,), 2.5, device=device)
    fn = df.p_sample_loop if p["sampler"] == "p" else df.ddim_sample_loop
    seam_kw = {"philox": dict(rng="philox", philox_seed=10), "torch": dict(rng="torch", progress=True),
               "stepwise": dict(rng="torch", fused=False)}[args.seam]
    if args.seam != "philox" and world > 1:
        raise SystemExit("--seam torch / stepwise draw from torch's generator: single GPU only (sharded runs use Philox)")

    def run(nsteps):
        """The last `nsteps` steps of the schedule (all of it when nsteps == loop length), for each sub-batch."""
        outs = []
        for s0 in range(0, B, sub):
            s1 = min(B, s0 + sub)
            y = {"seed": seed_d[s0:s1], "mfcc": mfcc_d[s0:s1]}
            if p["cfg"]:
                y["scale"] = scale_d[s0:s1]
            out, left = None, nsteps
            while left > 0:
                n = min(left, loop_steps)
                kw = dict(seam_kw)
                if args.seam == "philox":
                    kw["sample_offset"] = lo + s0
                out = fn(model, (s1 - s0, J, 1, T), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=loop_steps - n, **kw)
                left -= n
            outs.append(out)
        local = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        full = dist_util.gather_samples(local, total)        # end-of-loop gather (RCCL over xGMI when N > 1)
        return full if full is not None else local

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: {p['label']}{' (custom overrides)' if custom else ''}: {B} samples on {device}, "
        f"warm-up {args.warmup} steps ...")
    if args.warmup > 0:
        run(min(args.warmup, loop_steps))
    barrier()
    log(f"timing {steps} steps ...")
    eng0 = inner._get_engine(device)
    if rank == 0:
        eng0.profile_begin(512)      # HIP events around the first 512 FFN-1 GEMM launches of the timed region
    t0 = time.perf_counter()
    out = run(steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all()
    # what the ranks saw: proves which backend moved the gather and that N distinct devices took part
    mine = {"rank": rank, "device": str(device), "name": torch.cuda.get_device_name(device),
            "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", ""))}
    seen = [mine]
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, mine)
    if rank == 0 and args.save_samples:
        torch.save(out.cpu(), args.save_samples)

    nsub = (B + sub - 1) // sub
    ms_per_step = elapsed * 1e3 / steps                  # one step over the rank's whole batch (all its sub-batches)
    log(f"{elapsed:.3f} s for {steps} steps = {ms_per_step:.3f} ms/step")
    frames_per_sec = total * T / (loop_steps * ms_per_step * 1e-3)

    if rank == 0:
        from gesturediffusion_amd.engine import GDX_CFG, GDX_COND
        eng = inner._get_engine(device)
        flops_step = eng.forward_flops(GDX_CFG if p["cfg"] else GDX_COND) * nsub      # prepared shape = one sub-batch
        if B % sub:
            flops_step *= B / (nsub * sub)
        # dominant kernel = the MFMA GEMM; its north-star instance is FFN linear1 (+bias+GELU):
        # algorithmic FLOPs per launch = 2 * rows * d * ff, timed with HIP events on the launch stream
        rows, d, ff = (2 if p["cfg"] else 1) * min(sub, B) * (T + 1), p["d"], 1024
        gemm_us, gemm_launches = eng.profile_end()
        gemm_flops = 2.0 * rows * d * ff
        f16 = p["dtype"] in ("fp16", "bf16")
        traffic, traffic_src = None, None
        # the PMC passes were taken at the preset's one-GPU shape: a global batch split over several ranks is another GEMM
        if os.path.exists(TRAFFIC_FILE) and not custom and not (p["global_batch"] and world > 1):
            ent = json.load(open(TRAFFIC_FILE)).get(args.config)     # PMC passes cannot run inside this process
            if ent:
                traffic, traffic_src = ent["hbm_bytes_per_launch"], ent["source"]
        achieved = gemm_flops / (gemm_us * 1e-6) / 1e12 if gemm_us else 0.0
        peak = F16_MFMA_PEAK_TFLOPS if f16 else F32_MFMA_PEAK_TFLOPS
        loop_name = f"{loop_steps}-step {'p_sample_loop' if p['sampler'] == 'p' else 'ddim_sample_loop'}"
        topo = "MDM_Old (V1 encoder-only topology)" if p["arch"] == "mdm_old" else "MDM (V2: local attention + RoPE)"
        per_gpu = f"{B}/GPU" + (f" (global {total} over {world} GPU{'s' if world > 1 else ''}" +
                                (f", run as {nsub} sub-batches of {sub}" if nsub > 1 else "") + ")" if p["global_batch"] else "")
        rec = {
            "metric": f"denoised motion frames/sec ({loop_name}{' + CFG' if p['cfg'] else ''}, B={total if p['global_batch'] else p['batch']}, T={T}, d={p['d']})",
            "value": round(frames_per_sec, 2), "unit": "frames/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": p["scaling"], "vs_baseline": None, "dtype": ("bf16 (fp32 accumulate)" if p["dtype"] == "bf16" else "f16 (fp32 accumulate)") if f16 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{p['label']}{' with custom overrides' if custom else ''}: {topo} J={J} d={p['d']} ff=1024 L={p['L']} H=4, "
                            f"{loop_name}{' + classifier-free guidance 2.5 (2 model rows per sample)' if p['cfg'] else ''}, "
                            f"batch {per_gpu} x {T} frames, random weights, "
                            f"{ {'philox': 'in-kernel Philox noise', 'torch': 'torch-generator noise through the fused loop (reference caller kwargs, progress=True)', 'stepwise': 'step-wise callable protocol (model(x, t, **kw) + one update per step)'}[args.seam]}",
                "global_batch": total, "frames": T, "parallelism": f"dp{world} (independent samples, end-of-loop gather)",
                "step": "one denoising step (forward + fused sampler update) over the rank's batch",
                "timed": (f"one complete {loop_steps}-step loop" if steps == loop_steps else
                          f"last {steps} steps of the {loop_steps}-step schedule; value normalised to one loop" if steps < loop_steps else
                          f"{steps} steps = {steps / loop_steps:g} loops; value normalised to one loop"),
            },
            "dist": {"backend": dist.get_backend() if world > 1 else None, "world": world,
                     "launcher": os.environ.get("GDX_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "single process"),
                     "gather": ("batch_isend_irecv into one root buffer" if world > 1 else "none (one rank)"),
                     "ranks": seen},
            "loop_seconds": round(ms_per_step * loop_steps * 1e-3, 3),
            "frame_steps_per_sec": round(total * T / (ms_per_step * 1e-3), 1),
            "step_tflops": round(world * flops_step / (ms_per_step * 1e-3) / 1e12, 2),
            ("step_frac_of_f16_mfma_peak" if f16 else "step_frac_of_f32_mfma_peak"):
                round(flops_step / (ms_per_step * 1e-3) / 1e12 / peak, 4),
            "roofline": {"bound": "mfma", "kernel": ("gemmh kernel (gemmh.hip)" if f16 else "gemm4_kernel (gemm2.hip)") +
                                                    ", FFN linear1 + bias + GELU, "
                                                    f"M={rows} N={ff} K={d}; HIP events on the launch stream around "
                                                    f"{gemm_launches} launches inside the timed loop",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "avg_launch_us": round(gemm_us, 2), "flops_per_launch": gemm_flops},
        }
        if not args.no_cpu_baseline and world == 1:
            nb = min(sub, B)
            rec["cpu_baseline"] = cpu_baseline(p, cfg, sd, nb, seedp[:nb], mfcc[:nb], loop_steps, args.cpu_seconds,
                                               full_loops=args.config == "1" and not custom)
            if nb != B:
                rec["cpu_baseline"]["sample"] += f" (one sub-batch of {nb}; frames/s does not depend on the number of sub-batches)"
            rec["gpu_over_cpu"] = round(frames_per_sec / world / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec), file=record_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
