"""Whole-loop parity at BASELINE config 2 shapes: python tools/full_loop_parity.py STEPS  (fused Philox loop on the GPU vs the CPU
oracle fed the same Philox noise; B = 2)."""
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from test_gpu_parity import _real_cfg, build_model, dev, _diffusion
from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs
from oracle import mdm_forward as omf, sampler as osamp, schedule as osch, philox
import os
def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)
torch.set_num_threads(usable_cores())
print("threads", usable_cores(), flush=True)
for arch, J, T, steps in (("mdm_old", 263, 196, int(sys.argv[1])), ("mdm", 263, 200, int(sys.argv[1]))):
    cfg = _real_cfg(arch, J, 512)
    sd = init_state_dict(cfg, seed=0)
    m = build_model(arch, cfg, sd)
    B = 2
    _, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=10)
    y = {"seed": seedp.to(dev()), "mfcc": mfcc.to(dev())}
    resp = [steps] if steps < 1000 else ""
    df = _diffusion(resp if resp else [1000])
    t0 = time.time()
    out = df.p_sample_loop(m, (B, J, 1, T), clip_denoised=False, model_kwargs={"y": y}, rng="philox", philox_seed=10).cpu()
    tg = time.time() - t0
    tab, tmap = osch.make_tables("cosine", 1000, resp if resp else [1000])
    n = tab.num_timesteps
    print(f"{arch}: GPU loop done in {tg:.1f} s", flush=True)
    tape = torch.stack([torch.from_numpy(philox.normal(B, J * T, 10, 0, k)).view(B, J, 1, T) for k in range(n + 1)])
    t0 = time.time()
    with torch.no_grad():
        want = osamp.sample_loop(lambda x, t, yy: omf.forward(sd, cfg, x, t, yy), tab, tmap, (B, J, 1, T), tape, {"seed": seedp, "mfcc": mfcc}, kind="p")
    tc = time.time() - t0
    err = float((out - want).abs().max() / want.abs().max())
    print(f"{arch} T={T} {n}-step ancestral loop, B={B}: GPU {tg:.1f} s, oracle {tc:.1f} s, rel err {err:.2e}, max|ref| {float(want.abs().max()):.3f}", flush=True)

# BASELINE config 3's loop: 100-step DDIM (eta 0) under classifier-free guidance 2.5, V1 topology, T = 196
from gesturediffusion_amd.model.cfg_sampler import ClassifierFreeSampleModel
arch, J, T, B = "mdm_old", 263, 196, 2
cfg = _real_cfg(arch, J, 512)
sd = init_state_dict(cfg, seed=0)
m = ClassifierFreeSampleModel(build_model(arch, cfg, sd))
_, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=10)
scale = torch.full((B,), 2.5)
y = {"seed": seedp.to(dev()), "mfcc": mfcc.to(dev()), "scale": scale.to(dev())}
out = _diffusion("ddim100").ddim_sample_loop(m, (B, J, 1, T), clip_denoised=False, model_kwargs={"y": y}, rng="philox", philox_seed=10).cpu()
tab, tmap = osch.make_tables("cosine", 1000, "ddim100")
tape = torch.stack([torch.from_numpy(philox.normal(B, J * T, 10, 0, k)).view(B, J, 1, T) for k in range(101)])
with torch.no_grad():
    want = osamp.sample_loop(lambda x, t, yy: omf.cfg_forward(sd, cfg, x, t, yy), tab, tmap, (B, J, 1, T), tape,
                             {"seed": seedp, "mfcc": mfcc, "scale": scale}, kind="ddim")
print(f"config-3 loop (100-step DDIM + CFG 2.5, {arch} T={T}, B={B}): rel err {float((out - want).abs().max() / want.abs().max()):.2e}", flush=True)

