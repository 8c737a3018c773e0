"""Seeding helper: drop-in for reference `utils/fixseed.py:6-10`."""
import random

import numpy as np
import torch


def fixseed(seed):
    """Seed Python's, NumPy's and torch's global generators (the loops draw x_T and z from torch's when rng="torch")."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    torch.backends.cudnn.benchmark = False                # MIOpen autotuning off, as the reference sets it
