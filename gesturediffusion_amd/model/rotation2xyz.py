"""Post-processing stub with the reference's observable behaviour on this path.

Reference `model/rotation2xyz.py:20-23`: `__call__` returns `x` for pose_rep == 'xyz' (the only
value `sample/generate.py:165-169` passes) and needs SMPL body-model files otherwise.  SMPL
fitting is out of scope, so other pose_reps raise.
"""
import torch.nn as nn


class Rotation2xyz:
    def __init__(self, device="cpu", dataset="amass"):
        self.device, self.dataset = device, dataset
        self.smpl_model = nn.Identity()

    def __call__(self, x, mask=None, pose_rep="xyz", **kwargs):
        if pose_rep == "xyz":
            return x
        raise NotImplementedError("SMPL-based rotation->xyz conversion is outside the sampling hot path")
