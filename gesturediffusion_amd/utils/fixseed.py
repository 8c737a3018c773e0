"""Seeding helper: drop-in for reference `utils/fixseed.py:6-10`."""
import random

import numpy as np
import torch


def fixseed(seed):
    torch.backends.cudnn.benchmark = False
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
