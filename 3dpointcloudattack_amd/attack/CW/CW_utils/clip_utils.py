"""MI355X mirror of attack/CW/CW_utils/clip_utils.py — each functor is ONE HIP launch (pc3d_clip_f32) instead of
the reference's 10-25 elementwise kernels. Inputs [B,3,K]; results are new tensors (no grad), like the reference."""
import torch
import torch.nn as nn

from .... import ops


def _f(t):
    return t.detach().float()


class ClipPointsL2(nn.Module):
    """clip_utils.py:5-29 — global L2 norm of the perturbation <= budget."""

    def __init__(self, budget):
        super(ClipPointsL2, self).__init__()
        self.budget = budget

    def forward(self, pc, ori_pc):
        with torch.no_grad():
            return ops.clip(_f(pc), _f(ori_pc), None, self.budget, mode="global")


class ClipPointsLinf(nn.Module):
    """clip_utils.py:32-56 — per-point L2 norm of the perturbation <= budget (named Linf in the reference)."""

    def __init__(self, budget):
        super(ClipPointsLinf, self).__init__()
        self.budget = budget

    def forward(self, pc, ori_pc):
        with torch.no_grad():
            return ops.clip(_f(pc), _f(ori_pc), None, self.budget, mode="point")


class ProjectInnerPoints(nn.Module):
    """clip_utils.py:59-108 — points moved against the normal are projected onto the tangent direction."""

    def __init__(self):
        super(ProjectInnerPoints, self).__init__()

    def forward(self, pc, ori_pc, normal=None):
        with torch.no_grad():
            if normal is None:
                return pc
            return ops.clip(_f(pc), _f(ori_pc), _f(normal), 0.0, mode="point")


class ProjectInnerClipLinf(nn.Module):
    """clip_utils.py:111-136 — projection then per-point clip, fused in one launch."""

    def __init__(self, budget):
        super(ProjectInnerClipLinf, self).__init__()
        self.project_inner = ProjectInnerPoints()
        self.clip_linf = ClipPointsLinf(budget=budget)
        self.budget = budget

    def forward(self, pc, ori_pc, normal=None):
        with torch.no_grad():
            return ops.clip(_f(pc), _f(ori_pc), None if normal is None else _f(normal), self.budget, mode="point")
