#!/usr/bin/env python3
"""RCCL sanity on one GPU (world size 1): the process-group options the multi-GPU path uses (`device_id`, barrier, all_reduce on
the device, all_gather_object) load and run with torch's nccl backend on this image.  Not a scaling test: the N > 1 gather
(batch_isend_irecv) needs N GPUs and is exercised only by the driver's 8-GPU run."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
x = torch.ones(1 << 20, device=dev, dtype=torch.float64)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
objs = [None]; dist.all_gather_object(objs, {"rank": 0, "name": torch.cuda.get_device_name(dev)})
from gesturediffusion_amd.utils import dist_util
full = dist_util.gather_samples(torch.randn(3, 5, 1, 7, device=dev), 3)
torch.cuda.synchronize()
print(f"rccl world-1 ok: backend={dist.get_backend()} {objs[0]} gather={tuple(full.shape)} in {time.time() - t0:.1f} s")
dist.destroy_process_group()
