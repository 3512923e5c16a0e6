"""MI355X mirror of attack/GeoA3/loss_utils.py — GeoA3's geometry-aware regularisers on [B,3,N] clouds.

Every neighbour search is a fused HIP launch (knn_utils.knn_points); none builds a [B,N,N] matrix. The functions that
the reference writes with explicit N x N broadcasts (displacement_loss, corresponding_normal_loss, repulsion_loss,
distance_kmean_loss) use the same kernels. Functions accept an optional pre-computed adv->ori nearest-neighbour
result (``nn_ao``) so the iteration computes that search ONCE instead of four times.
"""
import math

import torch

from ...model import pointnet2_utils
from .knn_utils import knn_gather, knn_points
from .utility import _normalize


def norm_l2_loss(adv_pc, ori_pc):
    """loss_utils.py:33-34."""
    return ((adv_pc - ori_pc) ** 2).sum(1).sum(1)


def _nn(adv_pc, ori_pc, nn_ao=None):
    return nn_ao if nn_ao is not None else knn_points(adv_pc.permute(0, 2, 1), ori_pc.permute(0, 2, 1), K=1)


def chamfer_loss(adv_pc, ori_pc, nn_ao=None):
    """:36-43 — mean NN squared distance, both directions, summed."""
    adv_KNN = _nn(adv_pc, ori_pc, nn_ao)
    ori_KNN = knn_points(ori_pc.permute(0, 2, 1), adv_pc.permute(0, 2, 1), K=1)
    return adv_KNN.dists.contiguous().squeeze(-1).mean(-1) + ori_KNN.dists.contiguous().squeeze(-1).mean(-1)


def pseudo_chamfer_loss(adv_pc, ori_pc, nn_ao=None):
    """:45-51 — adv -> ori side only."""
    return _nn(adv_pc, ori_pc, nn_ao).dists.contiguous().squeeze(-1).mean(-1)


def hausdorff_loss(adv_pc, ori_pc, nn_ao=None):
    """:53-58 — one-sided max of the adv -> ori NN squared distances."""
    return _nn(adv_pc, ori_pc, nn_ao).dists.contiguous().squeeze(-1).max(-1)[0]


def _kappa(pc, normal, k):
    """mean_j |<normalize(p_j - p_i), n_i>| over the k nearest neighbours (:63-70, :83-90). On the GPU: one neighbour
    search + one launch (pc3d_kappa_f32, with its own backward) instead of the [b,3,n,k] tensors; the step-by-step
    form below is kept for normals that need a gradient."""
    if pc.is_cuda and not normal.requires_grad and pc.dtype == torch.float32 and normal.dtype == torch.float32:
        from ... import ops
        pd = pc.detach()
        idx = ops.knn_raw(pd, pd, k + 1, q_cf=True, r_cf=True)[1]
        return ops.kappa(pc, normal.detach(), idx, cf=True)
    pts = pc.permute(0, 2, 1).contiguous()
    inter_KNN = knn_points(pts, pts, K=k + 1)
    nn_pts = knn_gather(pts, inter_KNN.idx).permute(0, 3, 1, 2)[:, :, :, 1:].contiguous()  # [b,3,n,k]
    vectors = _normalize(nn_pts - pc.unsqueeze(3))
    return torch.abs((vectors * normal.unsqueeze(3)).sum(1)).mean(2)  # [b,n]


def _get_kappa_ori(pc, normal, k=2):
    """:60-70."""
    return _kappa(pc, normal, k)


def _get_kappa_adv(adv_pc, ori_pc, ori_normal, k=2, nn_ao=None, knn_idx=None):
    """:72-90 — normals of the nearest ori point, curvature proxy from the adv cloud's own k-NN. knn_idx: int32 [B,N,k+1]
    neighbour lists of adv_pc (self first) when the caller already has them (the victim's own input graph,
    graphed.input_knn), else they are searched here."""
    intra_KNN = _nn(adv_pc, ori_pc, nn_ao)
    if adv_pc.is_cuda and adv_pc.dtype == torch.float32 and ori_normal.dtype == torch.float32 and not ori_normal.requires_grad:
        # one neighbour search + one launch: the normal gather happens inside the curvature kernel (pc3d_kappa_gather_f32)
        from ... import ops
        pd = adv_pc.detach()
        idx = knn_idx if knn_idx is not None else ops.knn_raw(pd, pd, k + 1, q_cf=True, r_cf=True)[1]
        return ops.kappa_gather(adv_pc, ori_normal, intra_KNN.idx.squeeze(-1).contiguous(), idx)
    normal = knn_gather(ori_normal.permute(0, 2, 1).contiguous(), intra_KNN.idx).permute(0, 3, 1, 2).squeeze(3).contiguous()
    return _kappa(adv_pc, normal, k), normal


def curvature_loss(adv_pc, ori_pc, adv_kappa, ori_kappa, k=2, nn_ao=None):
    """:92-105."""
    intra_KNN = _nn(adv_pc, ori_pc, nn_ao)
    onenn_ori_kappa = torch.gather(ori_kappa, 1, intra_KNN.idx.squeeze(-1)).contiguous()
    return ((adv_kappa - onenn_ori_kappa) ** 2).mean(-1)


def displacement_loss(adv_pc, ori_pc, k=16):
    """:107-115."""
    b, _, n = adv_pc.size()
    with torch.no_grad():
        pts = ori_pc.permute(0, 2, 1).contiguous()
        inter_idx = knn_points(pts, pts, K=k + 1).idx[:, :, 1:].contiguous()
    theta_distance = ((adv_pc - ori_pc) ** 2).sum(1)
    nn_theta = torch.gather(theta_distance, 1, inter_idx.view(b, n * k)).view(b, n, k)
    return ((nn_theta - theta_distance.unsqueeze(2)) ** 2).mean(2)


def corresponding_normal_loss(adv_pc, normal, k=2):
    """:117-125."""
    return _kappa(adv_pc, normal, k)


def repulsion_loss(pc, k=4, h=0.03):
    """:127-131."""
    pts = pc.permute(0, 2, 1).contiguous()
    dis = knn_points(pts, pts, K=k + 1).dists[:, :, 1:].contiguous()
    return -(dis * torch.exp(-(dis ** 2) / (h ** 2))).mean(2)


def distance_kmean_loss(pc, k):
    """:133-141 (the reference adds 1e-12 inside the square root's argument difference; kept on the distances)."""
    b, _, n = pc.size()
    pts = pc.permute(0, 2, 1).contiguous()
    res = knn_points(pts, pts, K=k + 1)
    dis = torch.sqrt(res.dists.clamp(min=0) + 3e-24)
    dis_mean = dis[:, :, 1:].contiguous().mean(-1)
    idx = res.idx[:, :, 1:].contiguous()
    dis_mean_k = torch.gather(dis_mean, 1, idx.view(b, n * k)).view(b, n, k)
    return torch.abs(dis_mean.unsqueeze(2) - dis_mean_k).mean(-1)


def kNN_smoothing_loss(adv_pc, k, threshold_coef=1.05):
    """:143-157."""
    pts = adv_pc.permute(0, 2, 1).contiguous()
    inter_KNN = knn_points(pts, pts, K=k + 1)
    knn_dis = inter_KNN.dists[:, :, 1:].contiguous().mean(-1)
    threshold = knn_dis.mean(-1) + threshold_coef * knn_dis.std(-1)
    condition = torch.gt(knn_dis, threshold.unsqueeze(1)).float()
    return (knn_dis * condition).mean(1)


def uniform_loss(adv_pc, percentages=[0.004, 0.006, 0.008, 0.010, 0.012], radius=1.0, k=2):
    """:159-197 — the reference calls furthest_point_sample / gather_operation / ball_query / grouping_operation on a
    module that does not define them (SURVEY A-12); with this package's FPS / ball-query kernels it is implementable.
    FPS starts at index 0 (the CUDA op those names come from is deterministic)."""
    if adv_pc.size(1) == 3:
        adv_pc = adv_pc.permute(0, 2, 1).contiguous()
    b, n, _ = adv_pc.size()
    npoint = int(n * 0.05)
    loss = None
    from ... import ops
    for p in percentages:
        p = p * 4
        nsample = int(n * p)
        r = math.sqrt(p * radius)
        disk_area = math.pi * (radius ** 2) * p / nsample
        expect_len = math.sqrt(disk_area)     # a host scalar: no per-call host->device copy
        fps_idx = ops.fps(adv_pc, npoint, None)
        new_xyz = pointnet2_utils.index_points(adv_pc, fps_idx)
        idx = ops.ball_query(r, nsample, adv_pc, new_xyz.detach())
        grouped = pointnet2_utils.index_points(adv_pc, idx).reshape(b * npoint, nsample, 3)
        uniform_dis = knn_points(grouped, grouped, K=k + 1).dists[:, :, 1:].contiguous()
        uniform_dis = torch.sqrt(torch.abs(uniform_dis) + 1e-12).mean(dim=-1)
        uniform_dis = ((uniform_dis - expect_len) ** 2 / (expect_len + 1e-12)).reshape(-1)
        mean = uniform_dis.mean() * math.pow(p * 100, 2)
        loss = mean if loss is None else loss + mean
    return loss / len(percentages)
