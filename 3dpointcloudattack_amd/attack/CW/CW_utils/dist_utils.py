"""MI355X mirror of attack/CW/CW_utils/dist_utils.py — the distance regularisers the attacks call every step.

Same class names / constructor arguments / forward signatures ``dist_func(adv_pc, ori_pc, weights=None,
batch_avg=True)``. The point-set work runs in fused HIP kernels (no [B,K,K] matrices):
ChamferDist/HausdorffDist -> pc3d_nn_bidir_f32 (+bwd), KNNDist -> pc3d_knn_f32 (+bwd).

Layout note (deliberate, documented in DESIGN.md): the reference documents [B,K,3] inputs, yet CW.attack hands its
dist_func [B,3,K] tensors (attack/CW/CW_attack.py:161). The Chamfer/Hausdorff/kNN functors here accept BOTH:
a tensor whose last dim is 3 is [B,K,3]; otherwise dim 1 must be 3 and it is read channel-first, zero-copy.
"""
import torch
import torch.nn as nn

from .... import ops
from .distance import chamfer, hausdorff  # noqa: F401  (reference re-exports these)


def _weights(weights, B, device):
    if weights is None:
        return torch.ones((B,), device=device)
    return weights.float().to(device)


def _is_cf(t):
    """channel-first [B,3,K]?  ([B,K,3] wins when both dims are 3, as documented by the reference)."""
    if t.dim() != 3:
        raise ValueError(f"expected a 3-D point tensor, got shape {tuple(t.shape)}")
    if t.shape[-1] == 3:
        return False
    if t.shape[1] == 3:
        return True
    raise ValueError(f"expected [B,K,3] or [B,3,K] points, got shape {tuple(t.shape)}")


class L2Dist(nn.Module):
    """dist_utils.py:9-35 — ||adv-ori||_F per sample * weights."""

    def __init__(self):
        super(L2Dist, self).__init__()

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        B = adv_pc.shape[0]
        weights = _weights(weights, B, adv_pc.device)
        dist = torch.sqrt(torch.sum((adv_pc - ori_pc) ** 2, dim=[1, 2]))  # [B]
        dist = dist * weights
        if batch_avg:
            return dist.mean()
        return dist


class _SetDist(nn.Module):
    _reduce = "mean"

    def __init__(self, method='adv2ori'):
        super(_SetDist, self).__init__()
        self.method = method

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        B = adv_pc.shape[0]
        weights = _weights(weights, B, adv_pc.device)
        a, o, a_cf, o_cf = adv_pc.float(), ori_pc.float(), _is_cf(adv_pc), _is_cf(ori_pc)
        if self.method == 'adv2ori':          # one search: the other direction is never looked at (:62-63)
            loss = ops.set_distance_one(a, o, self._reduce, a_cf=a_cf, b_cf=o_cf)
        elif self.method == 'ori2adv':
            loss = ops.set_distance_one(o, a, self._reduce, a_cf=o_cf, b_cf=a_cf)
        else:
            loss1, loss2 = ops.set_distance(a, o, self._reduce, a_cf=a_cf, b_cf=o_cf)  # adv2ori, ori2adv
            loss = (loss1 + loss2) / 2.
        loss = loss * weights
        if batch_avg:
            return loss.mean()
        return loss

    def per_sample_terms(self, adv_pc, ori_pc, up=1.0, mean=True):
        """The functor's per-sample values as a list of [B] autograd tensors whose backward uses a gradient FIXED here —
        d loss / d term_b for a loss that contains `up * forward(adv_pc, ori_pc)` (batch mean; `up` a float, or a [B]
        tensor of per-sample weights standing for forward's `weights`) — and ignores the incoming one. mean=False: `up`
        already is d loss / d term_b (the caller divided by the batch size). An attack loop calls
        torch.autograd.backward on the terms directly: none of the weight / mean / scale launches around the functor,
        forward or backward (ops.scaled_gvec keeps autograd's arithmetic order)."""
        B = adv_pc.shape[0]
        a, o, a_cf, o_cf = adv_pc.float(), ori_pc.float(), _is_cf(adv_pc), _is_cf(ori_pc)
        if self.method == 'adv2ori':
            return [ops.set_distance_one(a, o, self._reduce, a_cf=a_cf, b_cf=o_cf, gvec=ops.scaled_gvec(up, 1.0, B, a.device, mean))]
        if self.method == 'ori2adv':
            return [ops.set_distance_one(o, a, self._reduce, a_cf=o_cf, b_cf=a_cf, gvec=ops.scaled_gvec(up, 1.0, B, a.device, mean))]
        l1, l2 = ops.set_distance(a, o, self._reduce, a_cf=a_cf, b_cf=o_cf, gvec=ops.scaled_gvec(up, 0.5, B, a.device, mean))
        return [l1, l2]


class ChamferDist(_SetDist):
    """dist_utils.py:38-72 — mean squared NN distance, method in {adv2ori, ori2adv, both}."""
    _reduce = "mean"


class HausdorffDist(_SetDist):
    """dist_utils.py:75-109 — max squared NN distance."""
    _reduce = "max"


class KNNDist(nn.Module):
    """dist_utils.py:112-160 — AAAI'20 kNN-distance outlier penalty (k=5, alpha=1.05)."""

    def __init__(self, k=5, alpha=1.05):
        super(KNNDist, self).__init__()
        self.k = k
        self.alpha = alpha

    def forward(self, pc, weights=None, batch_avg=True):
        B = pc.shape[0]
        cf = _is_cf(pc)
        pc = pc.float()
        # search + one loss launch (mean neighbour distance, mean + alpha * std threshold, masked mean: :133-151)
        loss = ops.knn_outlier_loss(pc, self.k, self.alpha, cf)  # [B]
        weights = _weights(weights, B, pc.device)
        loss = loss * weights
        if batch_avg:
            return loss.mean()
        return loss

    def per_sample_terms(self, pc, up=1.0, mean=True):
        """As _SetDist.per_sample_terms, for a loss that contains `up * forward(pc)`."""
        B = pc.shape[0]
        return [ops.knn_outlier_loss(pc.float(), self.k, self.alpha, _is_cf(pc), gvec=ops.scaled_gvec(up, 1.0, B, pc.device, mean))]


class ClipPointsLinf(nn.Module):
    """dist_utils.py:162-186 — duplicate of clip_utils.ClipPointsLinf kept for import compatibility."""

    def __init__(self, budget):
        super(ClipPointsLinf, self).__init__()
        self.budget = budget

    def forward(self, pc, ori_pc):
        with torch.no_grad():
            return ops.clip(pc.detach().float(), ori_pc.detach().float(), None, self.budget, mode="point")


class ChamferkNNDist(nn.Module):
    """dist_utils.py:189-223 — chamfer_weight*ChamferDist + knn_weight*KNNDist (5 / 3 by default)."""

    def __init__(self, chamfer_method='adv2ori', knn_k=5, knn_alpha=1.05, chamfer_weight=5., knn_weight=3.):
        super(ChamferkNNDist, self).__init__()
        self.chamfer_dist = ChamferDist(method=chamfer_method)
        self.knn_dist = KNNDist(k=knn_k, alpha=knn_alpha)
        self.w1 = chamfer_weight
        self.w2 = knn_weight

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        chamfer_loss = self.chamfer_dist(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg)
        knn_loss = self.knn_dist(adv_pc, weights=weights, batch_avg=batch_avg)
        return chamfer_loss * self.w1 + knn_loss * self.w2

    def per_sample_terms(self, adv_pc, ori_pc, up=1.0, mean=True):
        """As _SetDist.per_sample_terms: the Chamfer terms, then the kNN term, the functor's two weights in their gradients."""
        if torch.is_tensor(up):
            up1, up2 = up * self.w1, up * self.w2
        else:
            import numpy as np
            up1, up2 = np.float32(up) * np.float32(self.w1), np.float32(up) * np.float32(self.w2)
        return self.chamfer_dist.per_sample_terms(adv_pc, ori_pc, up1, mean) + self.knn_dist.per_sample_terms(adv_pc, up2, mean)


class FarthestDist(nn.Module):
    """dist_utils.py:226-254 — farthest intra-cluster pair, summed over added clusters (small tensors)."""

    def __init__(self):
        super(FarthestDist, self).__init__()

    def forward(self, adv_pc, weights=None, batch_avg=True):
        B = adv_pc.shape[0]
        weights = _weights(weights, B, adv_pc.device)
        delta_matrix = adv_pc[:, :, None, :, :] - adv_pc[:, :, :, None, :] + 1e-7
        norm_matrix = torch.norm(delta_matrix, p=2, dim=-1)  # [B, na, np, np]
        max_matrix = torch.max(norm_matrix, dim=2)[0]
        far_dist = torch.max(max_matrix, dim=2)[0]  # [B, num_add]
        far_dist = torch.sum(far_dist, dim=1)  # [B]
        loss = far_dist * weights
        if batch_avg:
            return loss.mean()
        return loss


class FarChamferDist(nn.Module):
    """dist_utils.py:257-292."""

    def __init__(self, num_add, chamfer_method='adv2ori', chamfer_weight=0.1):
        super(FarChamferDist, self).__init__()
        self.num_add = num_add
        self.far_dist = FarthestDist()
        self.chamfer_dist = ChamferDist(method=chamfer_method)
        self.cd_w = chamfer_weight

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        B = adv_pc.shape[0]
        chamfer_loss = self.chamfer_dist(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg)
        adv_clusters = adv_pc.view(B, self.num_add, -1, 3)
        far_loss = self.far_dist(adv_clusters, weights=weights, batch_avg=batch_avg)
        return far_loss + chamfer_loss * self.cd_w


class L2ChamferDist(nn.Module):
    """dist_utils.py:295-333."""

    def __init__(self, num_add, chamfer_method='adv2ori', chamfer_weight=0.2):
        super(L2ChamferDist, self).__init__()
        self.num_add = num_add
        self.chamfer_dist = ChamferDist(method=chamfer_method)
        self.cd_w = chamfer_weight
        self.l2_dist = L2Dist()

    def forward(self, adv_pc, ori_pc, adv_obj, ori_obj, weights=None, batch_avg=True):
        B = adv_pc.shape[0]
        chamfer_loss = self.chamfer_dist(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg)
        l2_loss = self.l2_dist(adv_obj.view(B, -1, 3), ori_obj.view(B, -1, 3), weights=weights, batch_avg=batch_avg)
        return l2_loss + self.cd_w * chamfer_loss
