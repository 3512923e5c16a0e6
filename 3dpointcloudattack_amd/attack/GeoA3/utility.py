"""MI355X mirror of the numerical helpers of attack/GeoA3/utility.py that the GeoA3 loop calls (normal estimation,
perpendicular jitter, FPS on [B,3,N], comparison and padding helpers). File IO / plotting / progress-bar utilities of
that file (:221-719) are outside the attack path and are not mirrored."""
import numpy as np
import torch

from ... import ops
from .knn_utils import knn_gather, knn_points


def _normalize(input, p=2, dim=1, eps=1e-12):
    """utility.py:33-34."""
    return input / input.norm(p, dim, keepdim=True).clamp(min=eps).expand_as(input)


def jitter_input(data, sigma=0.01, clip=0.05):
    """utility.py:36-41."""
    assert data.size(1) == 3
    assert (clip > 0)
    B, _, N = data.size()
    return ops.h2d(torch.clamp(sigma * torch.randn(B, 3, N), -1 * clip, clip), data.device)


def _nbr_cov(pc, k):
    """k nearest neighbours (self excluded) of every point and their 3x3 covariance, batched: [B,n,3,k], [B,n,3,3]."""
    b, _, n = pc.size()
    pts = pc.permute(0, 2, 1).contiguous()
    inter_KNN = knn_points(pts, pts, K=k + 1)
    nn_pts = knn_gather(pts, inter_KNN.idx)[:, :, 1:, :].permute(0, 1, 3, 2)       # [b,n,3,k]
    centred = nn_pts - torch.mean(nn_pts, dim=3, keepdim=True)
    cov = (1.0 / (k - 1)) * torch.matmul(centred, centred.transpose(2, 3))          # [b,n,3,3]
    return centred, cov


def estimate_normal(pc, k):
    """utility.py:43-92 — normal = eigenvector of the smallest eigenvalue of the k-NN covariance, sign fixed against
    the neighbour direction. pc [b,3,n] -> [b,3,n]. The reference's batched `torch.symeig` of [n,3,3] matrices is a
    closed-form 3 x 3 eigen-solve in one launch here (pc3d_estimate_normal_f32; rocsolver took 5 ms per call)."""
    with torch.no_grad():
        pts = pc.float().permute(0, 2, 1).contiguous()
        idx = knn_points(pts, pts, K=k + 1).idx.to(torch.int32).contiguous()
        return ops.estimate_normal(pts, idx).permute(0, 2, 1).contiguous()


def estimate_normal_via_ori_normal(pc_adv, pc_ori, normal_ori, k):
    """utility.py:94-111."""
    intra_KNN = knn_points(pc_adv.permute(0, 2, 1), pc_ori.permute(0, 2, 1), K=k)
    inter_value = intra_KNN.dists[:, :, 0].contiguous()
    normal_pts = knn_gather(normal_ori.permute(0, 2, 1), intra_KNN.idx).permute(0, 3, 1, 2).contiguous()
    normal_pts_avg = normal_pts.mean(dim=-1)
    normal_pts_avg = normal_pts_avg / (normal_pts_avg.norm(dim=1) + 1e-12)
    normal_ori_select = normal_pts[:, :, :, 0]
    condition = (inter_value < 1e-6).unsqueeze(1).expand_as(normal_ori_select)
    return torch.where(condition, normal_ori_select, normal_pts_avg)


def get_perpendicular_jitter(vector, sigma=0.01, clip=0.05):
    """utility.py:113-117."""
    b, _, n = vector.size()
    aux_vector1 = ops.h2d(sigma * torch.randn(b, 3, n), vector.device)
    aux_vector2 = ops.h2d(sigma * torch.randn(b, 3, n), vector.device)
    return torch.clamp(torch.cross(vector, aux_vector1, dim=1), -1 * clip, clip) + \
        torch.clamp(torch.cross(vector, aux_vector2, dim=1), -1 * clip, clip)


def estimate_perpendicular(pc, k, sigma=0.01, clip=0.05):
    """utility.py:119-154 — random jitter in the two dominant local directions."""
    with torch.no_grad():
        b, _, n = pc.size()
        _, cov = _nbr_cov(pc.float(), k)
        eigenvalue, eigenvector = torch.linalg.eigh(cov)
        larger = torch.topk(eigenvalue, 2, dim=2, largest=True, sorted=False)[1]     # [b,n,2]
        v1 = torch.gather(eigenvector, 3, larger[:, :, 0][:, :, None, None].expand(b, n, 3, 1)).squeeze(3).permute(0, 2, 1)
        v2 = torch.gather(eigenvector, 3, larger[:, :, 1][:, :, None, None].expand(b, n, 3, 1)).squeeze(3).permute(0, 2, 1)
        aux1 = ops.h2d(sigma * torch.randn(b, n).unsqueeze(1), pc.device)
        aux2 = ops.h2d(sigma * torch.randn(b, n).unsqueeze(1), pc.device)
    return torch.clamp(v1 * aux1, -1 * clip, clip) + torch.clamp(v2 * aux2, -1 * clip, clip)


def _compare(output, target, gt, targeted):
    """utility.py:156-160."""
    if targeted:
        return output == target
    return output != gt


def farthest_points_sample(obj_points, num_points):
    """utility.py:178-190 — [b,3,n] -> [b,3,num_points]; start index from torch.randint like the reference. The
    reference compares Euclidean norms; FPS on squared distances selects the same points (sqrt is monotone)."""
    assert obj_points.size(1) == 3
    b, _, n = obj_points.size()
    start = ops.h2d(torch.randint(n, [b, 1]).view(-1), obj_points.device, torch.int32)
    sel = ops.fps(obj_points.float(), num_points, start, cf=True).long()
    return torch.gather(obj_points, 2, sel.unsqueeze(1).expand(b, 3, num_points))


def pad_larger_tensor_with_index_batch(small_verts, small_in_larger_idx_list, larger_tensor_shape):
    """utility.py:214-219."""
    b, _, n = small_verts.size()
    full = torch.zeros(b, 3, larger_tensor_shape, device=small_verts.device)
    for i in range(b):
        full[i, :, small_in_larger_idx_list[i][0][1:]] = small_verts[i]
    return full
