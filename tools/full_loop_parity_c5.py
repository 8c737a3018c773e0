"""Whole-loop parity of the reduced-precision modes at BASELINE config 5's shape (V2, J=498, d=1024, T=520, B=1):
python tools/full_loop_parity_c5.py STEPS   (fused Philox loop in fp32 / fp16 / bf16 vs the CPU oracle on the same Philox stream)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from test_gpu_parity import _real_cfg, build_model, dev, _diffusion
from gesturediffusion_amd.utils.init import init_state_dict, synthetic_inputs
from oracle import mdm_forward as omf, sampler as osamp, schedule as osch, philox


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
steps = int(sys.argv[1])
arch, J, T, B = "mdm", 498, 520, 1
cfg = _real_cfg(arch, J, 1024)
sd = init_state_dict(cfg, seed=0)
_, seedp, mfcc = synthetic_inputs(cfg, B, T, seed=10)
y = {"seed": seedp.to(dev()), "mfcc": mfcc.to(dev())}
outs = {}
for dt in ("fp32", "fp16", "bf16"):
    m = build_model(arch, cfg, sd)
    m.compute_dtype = dt
    outs[dt] = _diffusion([steps]).p_sample_loop(m, (B, J, 1, T), clip_denoised=False, model_kwargs={"y": y}, rng="philox", philox_seed=10).cpu()
    print(dt, "GPU loop done", flush=True)
tab, tmap = osch.make_tables("cosine", 1000, [steps])
tape = torch.stack([torch.from_numpy(philox.normal(B, J * T, 10, 0, k)).view(B, J, 1, T) for k in range(steps + 1)])
t0 = time.time()
img = tape[0]
mt = torch.tensor(tmap, dtype=torch.long)
with torch.no_grad():
    for k, i in enumerate(range(steps - 1, -1, -1)):
        t = torch.tensor([i] * B)
        x0 = omf.forward(sd, cfg, img, mt[t], {"seed": seedp, "mfcc": mfcc})
        img = osamp.p_sample_step(tab, x0, img, t, tape[1 + k])
        if k % 10 == 0:
            print(f"oracle step {k}/{steps} {time.time() - t0:.0f} s", flush=True)
for dt, o in outs.items():
    print(f"config-5 shape, {steps}-step ancestral loop, B={B}, {dt}: rel err vs oracle {float((o - img).abs().max() / img.abs().max()):.2e}", flush=True)
