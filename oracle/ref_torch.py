"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product package (3dpointcloudattack_amd/).

CPU restatement (plain torch ops, any device torch supports but used on CPU) of the reference's victim model,
attack functors and attack loops, written from the reference's behaviour with file:line anchors. Used by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only, as the checker / the timed CPU baseline.

Pinned: tests/golden/make_golden.py runs the REAL reference (imported read-only from /root/reference under the
no-op-.cuda() harness shim described in SURVEY §8(c)) on seeded weights/clouds and stores its outputs;
tests/test_oracle_golden.py compares this file against those vectors.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------------
# Victim model: PointNet (model/pointnet.py:14-48, 89-128, 130-148), unfused Conv1d / BatchNorm1d / max
# ----------------------------------------------------------------------------------------------------------
class STN3d(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 9)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = F.relu(self.bn3(self.conv3(x)))
        x = torch.max(x, 2, keepdim=True)[0].view(-1, 1024)
        x = F.relu(self.bn4(self.fc1(x)))
        x = F.relu(self.bn5(self.fc2(x)))
        x = self.fc3(x)
        iden = torch.eye(3, dtype=x.dtype, device=x.device).view(1, 9)
        return (x + iden).view(-1, 3, 3)


class PointNetfeat(nn.Module):
    def __init__(self, global_feat=True, feature_transform=False):
        super().__init__()
        assert global_feat and not feature_transform
        self.stn = STN3d()
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)

    def forward(self, x):
        trans = self.stn(x)
        x = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        x = torch.max(x, 2, keepdim=True)[0].view(-1, 1024)
        return x, trans, None


class PointNetCls(nn.Module):
    def __init__(self, k=2, feature_transform=False):
        super().__init__()
        self.feat = PointNetfeat(True, feature_transform)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k)
        self.dropout = nn.Dropout(p=0.3)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()

    def forward(self, x):
        x, trans, trans_feat = self.feat(x)
        x = F.relu(self.bn1(self.fc1(x)))
        x = F.relu(self.bn2(self.dropout(self.fc2(x))))
        x = self.fc3(x)
        return F.log_softmax(x, dim=1), trans, trans_feat


# ----------------------------------------------------------------------------------------------------------
# Functors (attack/CW/CW_utils/*)
# ----------------------------------------------------------------------------------------------------------
def batch_pairwise_dist(x, y):
    """distance.py:15-32 — |x|^2 + |y|^2 - 2xy through three bmm (the reference's own arithmetic)."""
    xx = torch.bmm(x, x.transpose(2, 1))
    yy = torch.bmm(y, y.transpose(2, 1))
    zz = torch.bmm(x, y.transpose(2, 1))
    rx = torch.diagonal(xx, dim1=1, dim2=2).unsqueeze(1).expand_as(zz.transpose(2, 1))
    ry = torch.diagonal(yy, dim1=1, dim2=2).unsqueeze(1).expand_as(zz)
    return rx.transpose(2, 1) + ry - 2 * zz


def chamfer(preds, gts):
    """distance.py:40-50."""
    P = batch_pairwise_dist(gts, preds)
    return torch.mean(torch.min(P, 1)[0], dim=1), torch.mean(torch.min(P, 2)[0], dim=1)


def hausdorff(preds, gts):
    """distance.py:58-70."""
    P = batch_pairwise_dist(gts, preds)
    return torch.max(torch.min(P, 1)[0], dim=1)[0], torch.max(torch.min(P, 2)[0], dim=1)[0]


def _w(weights, B, ref):
    return torch.ones((B,), dtype=torch.float32) if weights is None else weights.float()


class L2Dist:
    """dist_utils.py:9-35."""

    def __call__(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        dist = torch.sqrt(torch.sum((adv_pc - ori_pc) ** 2, dim=[1, 2])) * _w(weights, adv_pc.shape[0], adv_pc)
        return dist.mean() if batch_avg else dist


class ChamferDist:
    """dist_utils.py:38-72; inputs [B,K,3]. `dtype=torch.float64` evaluates the SAME expansion arithmetic in double
    (then rounds the losses to fp32): the reference's fp32 |x|^2+|y|^2-2xy has ~1 ulp-of-|x| noise, which for
    adv ~ ori (the start of every attack) is as large as the true gradient 2(x-y) and is amplified to full
    lr-sized steps by Adam's normalisation — so fp32-reference trajectories are not reproducible by ANY other
    arithmetic (measured: 29% of coordinates off by up to 0.03 after 24 steps). The double variant is the
    reference's algorithm without that noise and is what the HIP kernels (exact direct-difference) must match."""

    def __init__(self, method='adv2ori', fn=chamfer, dtype=None):
        self.method, self.fn, self.dtype = method, fn, dtype

    def __call__(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        if self.dtype is not None:
            l1, l2 = self.fn(adv_pc.to(self.dtype), ori_pc.to(self.dtype))
            l1, l2 = l1.float(), l2.float()
        else:
            l1, l2 = self.fn(adv_pc, ori_pc)
        loss = l1 if self.method == 'adv2ori' else (l2 if self.method == 'ori2adv' else (l1 + l2) / 2.)
        loss = loss * _w(weights, adv_pc.shape[0], adv_pc)
        return loss.mean() if batch_avg else loss


class HausdorffDist(ChamferDist):
    """dist_utils.py:75-109."""

    def __init__(self, method='adv2ori', dtype=None):
        super().__init__(method, hausdorff, dtype)


class KNNDist:
    """dist_utils.py:112-160."""

    def __init__(self, k=5, alpha=1.05):
        self.k, self.alpha = k, alpha

    def __call__(self, pc, weights=None, batch_avg=True):
        B, K = pc.shape[:2]
        pc = pc.transpose(2, 1)
        inner = -2. * torch.matmul(pc.transpose(2, 1), pc)
        xx = torch.sum(pc ** 2, dim=1, keepdim=True)
        dist = xx + inner + xx.transpose(2, 1)
        neg_value, _ = (-dist).topk(k=self.k + 1, dim=-1)
        value = torch.mean(-(neg_value[..., 1:]), dim=-1)
        with torch.no_grad():
            threshold = torch.mean(value, dim=-1) + self.alpha * torch.std(value, dim=-1)
            mask = (value > threshold[:, None]).float()
        loss = torch.mean(value * mask, dim=1) * _w(weights, B, pc)
        return loss.mean() if batch_avg else loss


class ChamferkNNDist:
    """dist_utils.py:189-223."""

    def __init__(self, chamfer_method='adv2ori', knn_k=5, knn_alpha=1.05, chamfer_weight=5., knn_weight=3.):
        self.cd, self.kd, self.w1, self.w2 = ChamferDist(chamfer_method), KNNDist(knn_k, knn_alpha), chamfer_weight, knn_weight

    def __call__(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        return self.cd(adv_pc, ori_pc, weights, batch_avg) * self.w1 + self.kd(adv_pc, weights, batch_avg) * self.w2


class ChannelFirst:
    """Adapter: the CW loop hands [B,3,K] tensors to dist_func (CW_attack.py:161); a point-set functor documented
    for [B,K,3] is applied on the transposed views, as attack/additional_exp/CW_attack.py:151-153 does."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        return self.fn(adv_pc.transpose(1, 2), ori_pc.transpose(1, 2), weights, batch_avg)


def _real_other(logits, targets):
    B, K = logits.shape
    one_hot = torch.zeros(B, K).scatter_(1, targets.view(-1, 1).long(), 1).float()
    real = torch.sum(one_hot * logits, dim=1)
    other = torch.max((1. - one_hot) * logits - one_hot * 10000., dim=1)[0]
    return real, other


class LogitsAdvLoss:
    """adv_utils.py:6-33."""

    def __init__(self, kappa=0.):
        self.kappa = kappa

    def __call__(self, logits, targets):
        real, other = _real_other(logits, targets)
        return torch.clamp(other - real + self.kappa, min=0.).mean()


class UntargetedLogitsAdvLoss(LogitsAdvLoss):
    """adv_utils.py:53-80."""

    def __call__(self, logits, targets):
        real, other = _real_other(logits, targets)
        return torch.clamp(real - other + self.kappa, min=0.).mean()


class CrossEntropyAdvLoss:
    """adv_utils.py:36-51."""

    def __call__(self, logits, targets):
        return F.nll_loss(logits, targets)


class ClipPointsLinf:
    """clip_utils.py:32-56."""

    def __init__(self, budget):
        self.budget = budget

    def __call__(self, pc, ori_pc):
        with torch.no_grad():
            diff = pc - ori_pc
            norm = torch.sum(diff ** 2, dim=1) ** 0.5
            scale = torch.clamp(self.budget / (norm + 1e-9), max=1.)
            return ori_pc + diff * scale[:, None, :]


class ClipPointsL2:
    """clip_utils.py:5-29."""

    def __init__(self, budget):
        self.budget = budget

    def __call__(self, pc, ori_pc):
        with torch.no_grad():
            diff = pc - ori_pc
            norm = torch.sum(diff ** 2, dim=[1, 2]) ** 0.5
            scale = torch.clamp(self.budget / (norm + 1e-9), max=1.)
            return ori_pc + diff * scale[:, None, None]


class ProjectInnerPoints:
    """clip_utils.py:59-108."""

    def __call__(self, pc, ori_pc, normal=None):
        with torch.no_grad():
            if normal is None:
                return pc
            diff = pc - ori_pc
            inner_mask = torch.sum(diff * normal, dim=1) < 0.
            vng = torch.cross(normal, diff, dim=1)
            vng_norm = torch.sum(vng ** 2, dim=1) ** 0.5
            vref = torch.cross(vng, normal, dim=1)
            vref_norm = torch.sum(vref ** 2, dim=1) ** 0.5
            diff_proj = diff * vref / (vref_norm[:, None, :] + 1e-9)
            opposite = (inner_mask & (vng_norm < 1e-6)).unsqueeze(1).expand_as(diff_proj)
            diff_proj = torch.where(opposite, torch.zeros_like(diff_proj), diff_proj)
            diff = torch.where(inner_mask.unsqueeze(1).expand_as(diff), diff_proj, diff)
            return ori_pc + diff


class ProjectInnerClipLinf:
    """clip_utils.py:111-136."""

    def __init__(self, budget):
        self.project, self.clip = ProjectInnerPoints(), ClipPointsLinf(budget)

    def __call__(self, pc, ori_pc, normal=None):
        return self.clip(self.project(pc, ori_pc, normal), ori_pc)


# ----------------------------------------------------------------------------------------------------------
# CW attack loop (attack/CW/CW_attack.py:57-260), vectorised over the batch exactly per sample
# ----------------------------------------------------------------------------------------------------------
def cw_attack(model, data, target, adv_func, dist_func, clip_func, attack_lr=1e-2, init_weight=10., max_weight=80.,
              binary_step=10, num_iter=500, attack_method="untarget", record=None):
    """Returns (o_bestdist [B] f64, o_bestattack [B,K,3] f64, success_num, last_pred). `record(step, it, adv)`
    is an optional per-iteration hook used to capture trajectories."""
    B, K = data.shape[:2]
    data = data.float().transpose(1, 2).contiguous()
    ori = data.clone().detach()
    target = target.long().view(-1)
    label_val = target.numpy().copy()
    lower = np.zeros((B,))
    upper = np.ones((B,)) * max_weight
    cur_w = np.ones((B,)) * init_weight
    o_bestdist = np.array([1e10] * B)
    o_bestscore = np.array([-1] * B)
    o_bestattack = np.zeros((B, 3, K))
    untarget = attack_method == 'untarget'
    input_val = pred_val = None
    for bstep in range(binary_step):
        adv = ori.clone().detach() + torch.randn((B, 3, K)) * 1e-7           # :94
        adv.requires_grad_()
        bestdist = np.array([1e10] * B)
        bestscore = np.array([-1] * B)
        opt = torch.optim.Adam([adv], lr=attack_lr, weight_decay=0.)          # :100
        for it in range(num_iter):
            logits = model(adv)[0]                                            # :115
            pred = torch.argmax(logits, dim=1)
            dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).detach().numpy()   # :129-131
            pred_val = pred.detach().numpy()
            input_val = adv.detach().numpy().copy()
            if record is not None:
                record(bstep, it, input_val)
            for e in range(B):                                                # :136-153
                ok = (pred_val[e] != label_val[e]) if untarget else (pred_val[e] == label_val[e])
                if dist_val[e] < bestdist[e] and ok:
                    bestdist[e], bestscore[e] = dist_val[e], pred_val[e]
                if dist_val[e] < o_bestdist[e] and ok:
                    o_bestdist[e], o_bestscore[e] = dist_val[e], pred_val[e]
                    o_bestattack[e] = input_val[e]
            adv_loss = adv_func(logits, target).mean()                        # :160
            dist_loss = dist_func(adv, ori, torch.from_numpy(cur_w)).mean()   # :161-163
            loss = adv_loss + dist_loss
            opt.zero_grad()
            loss.backward()
            opt.step()
            if clip_func is not None:                                         # :172-174
                adv.data = clip_func(adv.clone().detach(), ori)
        for e in range(B):                                                    # :182-200
            if untarget:
                ok = bestscore[e] != label_val[e] and bestscore[e] != -1 and bestdist[e] <= o_bestdist[e]
            else:
                ok = bestscore[e] == label_val[e] and bestscore[e] != -1 and bestdist[e] <= o_bestdist[e]
            if ok:
                lower[e] = max(lower[e], cur_w[e])
            else:
                upper[e] = min(upper[e], cur_w[e])
            cur_w[e] = (lower[e] + upper[e]) / 2.
    succ = (pred_val != label_val) if untarget else (pred_val == label_val)
    fail_idx = (lower == 0.)                                                  # :208-209
    o_bestattack[fail_idx] = input_val[fail_idx]
    return o_bestdist, o_bestattack.transpose((0, 2, 1)), int(succ.sum()), pred_val


# ----------------------------------------------------------------------------------------------------------
# seeded weights shared by the fixtures, the oracle and the HIP build (no checkpoint ships with the reference)
# ----------------------------------------------------------------------------------------------------------
# The recipe itself lives with the package (3dpointcloudattack_amd/seeding.py: bench.py and user code may need it, and
# the product never imports oracle/); re-exported here for the fixtures and the tests.
import importlib as _importlib

_seeding = _importlib.import_module("3dpointcloudattack_amd.seeding")
seeded_state_dict, state_sha256 = _seeding.seeded_state_dict, _seeding.state_sha256


# ----------------------------------------------------------------------------------------------------------
# PointNet++ ops and classifiers (model/pointnet2_utils.py, model/pointnet2_SSG.py, model/pointnet2_MSG.py)
# ----------------------------------------------------------------------------------------------------------
def square_distance(src, dst):
    """pointnet2_utils.py:19-38 (the reference's -2ab + a^2 + b^2 arithmetic)."""
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


def index_points(points, idx):
    """:41-57."""
    B = points.shape[0]
    view_shape = list(idx.shape)
    view_shape[1:] = [1] * (len(view_shape) - 1)
    repeat_shape = list(idx.shape)
    repeat_shape[0] = 1
    batch_indices = torch.arange(B, dtype=torch.long).view(view_shape).repeat(repeat_shape)
    return points[batch_indices, idx, :]


def farthest_point_sample(xyz, npoint, start=None):
    """:60-81; start=None draws torch.randint from the global generator exactly like the reference (:72)."""
    B, N, C = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N) * 1e10
    farthest = torch.randint(0, N, (B,), dtype=torch.long) if start is None else start.long().clone()
    batch_indices = torch.arange(B, dtype=torch.long)
    for i in range(npoint):
        centroids[:, i] = farthest
        centroid = xyz[batch_indices, farthest, :].view(B, 1, 3)
        dist = torch.sum((xyz - centroid) ** 2, -1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = torch.max(distance, -1)[1]
    return centroids


def query_ball_point(radius, nsample, xyz, new_xyz, exact=False):
    """:84-104. exact=True evaluates the distances in the direct-difference form (what the HIP kernel does) instead
    of the reference's expansion; the two differ only for points within fp32 rounding of the ball's surface."""
    B, N, C = xyz.shape
    _, S, _ = new_xyz.shape
    group_idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat([B, S, 1])
    if exact:
        d = new_xyz[:, :, None, :] - xyz[:, None, :, :]
        sqrdists = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    else:
        sqrdists = square_distance(new_xyz, xyz)
    group_idx[sqrdists > radius ** 2] = N
    group_idx = group_idx.sort(dim=-1)[0][:, :, :nsample]
    group_first = group_idx[:, :, 0].view(B, S, 1).repeat([1, 1, nsample])
    mask = group_idx == N
    group_idx[mask] = group_first[mask]
    return group_idx


def sample_and_group(npoint, radius, nsample, xyz, points, exact=False):
    """:107-135."""
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz, exact)
    grouped_xyz_norm = index_points(xyz, idx) - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1)
    else:
        new_points = grouped_xyz_norm
    return new_xyz, new_points


class PointNetSetAbstraction(nn.Module):
    """:158-199."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, exact=False):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all, self.exact = npoint, radius, nsample, group_all, exact
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for oc in mlp:
            self.mlp_convs.append(nn.Conv2d(last, oc, 1))
            self.mlp_bns.append(nn.BatchNorm2d(oc))
            last = oc

    def forward(self, xyz, points):
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        if self.group_all:
            B, N, C = xyz.shape
            new_xyz = torch.zeros(B, 1, C)
            g = xyz.view(B, 1, N, C)
            new_points = torch.cat([g, points.view(B, 1, N, -1)], dim=-1) if points is not None else g
        else:
            new_xyz, new_points = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points, self.exact)
        new_points = new_points.permute(0, 3, 2, 1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_points = F.relu(bn(conv(new_points)))
        new_points = torch.max(new_points, 2)[0]
        return new_xyz.permute(0, 2, 1), new_points


class PointNetSetAbstractionMsg(nn.Module):
    """:202-259."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, exact=False):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list, self.exact = npoint, radius_list, nsample_list, exact
        self.conv_blocks, self.bn_blocks = nn.ModuleList(), nn.ModuleList()
        for mlp in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last = in_channel + 3
            for oc in mlp:
                convs.append(nn.Conv2d(last, oc, 1))
                bns.append(nn.BatchNorm2d(oc))
                last = oc
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points):
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        B, N, C = xyz.shape
        S = self.npoint
        new_xyz = index_points(xyz, farthest_point_sample(xyz, S))
        outs = []
        for i, radius in enumerate(self.radius_list):
            group_idx = query_ball_point(radius, self.nsample_list[i], xyz, new_xyz, self.exact)
            grouped_xyz = index_points(xyz, group_idx) - new_xyz.view(B, S, 1, C)
            g = torch.cat([index_points(points, group_idx), grouped_xyz], dim=-1) if points is not None else grouped_xyz
            g = g.permute(0, 3, 2, 1)
            for conv, bn in zip(self.conv_blocks[i], self.bn_blocks[i]):
                g = F.relu(bn(conv(g)))
            outs.append(torch.max(g, 2)[0])
        return new_xyz.permute(0, 2, 1), torch.cat(outs, dim=1)


class PointNet_Ssg(nn.Module):
    """pointnet2_SSG.py:230-254."""

    def __init__(self, num_classes=40, exact=False):
        super().__init__()
        self.sa1 = PointNetSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False, exact)
        self.sa2 = PointNetSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False, exact)
        self.sa3 = PointNetSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True, exact)
        self.fc1, self.bn1, self.drop1 = nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.Dropout(0.4)
        self.fc2, self.bn2, self.drop2 = nn.Linear(512, 256), nn.BatchNorm1d(256), nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, num_classes)

    def forward(self, xyz):
        B = xyz.shape[0]
        l1_xyz, l1_points = self.sa1(xyz, None)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        x = l3_points.view(B, 1024)
        x = self.drop1(F.relu(self.bn1(self.fc1(x))))
        x = self.drop2(F.relu(self.bn2(self.fc2(x))))
        x = F.log_softmax(self.fc3(x), -1)
        return x, x, x


class PointNet_Msg(nn.Module):
    """pointnet2_MSG.py:194-226 (normal_channel=False as the attack drivers build it)."""

    def __init__(self, num_class, normal_channel=False, exact=False):
        super().__init__()
        assert not normal_channel
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0, [[32, 32, 64], [64, 64, 128], [64, 96, 128]], exact)
        self.sa2 = PointNetSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320, [[64, 64, 128], [128, 128, 256], [128, 128, 256]], exact)
        self.sa3 = PointNetSetAbstraction(None, None, None, 640 + 3, [256, 512, 1024], True, exact)
        self.fc1, self.bn1, self.drop1 = nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.Dropout(0.4)
        self.fc2, self.bn2, self.drop2 = nn.Linear(512, 256), nn.BatchNorm1d(256), nn.Dropout(0.5)
        self.fc3 = nn.Linear(256, num_class)

    def forward(self, xyz):
        B = xyz.shape[0]
        l1_xyz, l1_points = self.sa1(xyz, None)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        x = l3_points.view(B, 1024)
        x = self.drop1(F.relu(self.bn1(self.fc1(x))))
        x = self.drop2(F.relu(self.bn2(self.fc2(x))))
        x = F.log_softmax(self.fc3(x), -1)
        return x, x, x


# ----------------------------------------------------------------------------------------------------------
# KNN attack loop (attack/KNN/KNN_attack.py:56-246)
# ----------------------------------------------------------------------------------------------------------
def knn_attack(model, data, target, adv_func, dist_func, clip_func, attack_lr=1e-3, num_iter=2500,
               attack_method='untarget', record=None):
    """Returns (adv [B,K,3] float32 numpy, success_num)."""
    B, K = data.shape[:2]
    data = data.float().transpose(1, 2).contiguous()
    ori = data.clone().detach()
    if ori.shape[1] == 3:
        normal = ori
    else:
        normal, ori = ori[:, 3:, :], ori[:, :3, :]
    with torch.no_grad():
        model(ori)                                                   # :76 clean forward (consumes RNG for PointNet++)
    target = target.long().view(-1)
    adv = ori.clone().detach() + torch.randn((B, 3, K)) * 1e-7      # :84-85
    adv.requires_grad_()
    opt = torch.optim.Adam([adv], lr=attack_lr, weight_decay=0.)
    for it in range(num_iter):
        logits = model(adv)
        logits = logits[0] if isinstance(logits, tuple) else logits
        if record is not None:
            record(it, adv.detach().numpy().copy())
        adv_loss = adv_func(logits, target).mean()
        dist_loss = dist_func(adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous()).mean() * K   # :121-123
        loss = adv_loss + dist_loss
        opt.zero_grad()
        loss.backward()
        opt.step()
        adv.data = clip_func(adv.clone().detach(), ori, normal)      # :136
    with torch.no_grad():
        logits = model(adv)
        logits = logits[0] if isinstance(logits, tuple) else logits
        pred = torch.argmax(logits, dim=-1)
        succ = (pred != target) if attack_method == 'untarget' else (pred == target)
    return adv.detach().transpose(1, 2).contiguous().numpy(), int(succ.sum())


# ----------------------------------------------------------------------------------------------------------
# DGCNN (model/dgcnn.py:194-227, 262-328)
# ----------------------------------------------------------------------------------------------------------
def dgcnn_knn(x, k):
    """:194-200 — top-k of the negative squared distance (expansion form), self included."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pairwise_distance = -xx - inner - xx.transpose(2, 1)
    return pairwise_distance.topk(k=k, dim=-1)[1]


def get_graph_feature(x, k=20, idx=None):
    """:203-227."""
    batch_size, num_dims, num_points = x.shape
    if idx is None:
        idx = dgcnn_knn(x, k=k)
    idx_base = torch.arange(0, batch_size).view(-1, 1, 1) * num_points
    idx = (idx + idx_base).view(-1)
    xt = x.transpose(2, 1).contiguous()
    feature = xt.view(batch_size * num_points, -1)[idx, :].view(batch_size, num_points, k, num_dims)
    xr = xt.view(batch_size, num_points, 1, num_dims).repeat(1, 1, k, 1)
    return torch.cat((feature - xr, xr), dim=3).permute(0, 3, 1, 2).contiguous()


class DGCNN(nn.Module):
    def __init__(self, args, output_channels=40):
        super().__init__()
        self.k = args.k
        self.bn1, self.bn2, self.bn3, self.bn4 = nn.BatchNorm2d(64), nn.BatchNorm2d(64), nn.BatchNorm2d(128), nn.BatchNorm2d(256)
        self.bn5 = nn.BatchNorm1d(args.emb_dims)
        self.conv1 = nn.Sequential(nn.Conv2d(6, 64, 1, bias=False), self.bn1, nn.LeakyReLU(0.2))
        self.conv2 = nn.Sequential(nn.Conv2d(128, 64, 1, bias=False), self.bn2, nn.LeakyReLU(0.2))
        self.conv3 = nn.Sequential(nn.Conv2d(128, 128, 1, bias=False), self.bn3, nn.LeakyReLU(0.2))
        self.conv4 = nn.Sequential(nn.Conv2d(256, 256, 1, bias=False), self.bn4, nn.LeakyReLU(0.2))
        self.conv5 = nn.Sequential(nn.Conv1d(512, args.emb_dims, 1, bias=False), self.bn5, nn.LeakyReLU(0.2))
        self.linear1 = nn.Linear(args.emb_dims * 2, 512, bias=False)
        self.bn6 = nn.BatchNorm1d(512)
        self.dp1 = nn.Dropout(p=args.dropout)
        self.linear2 = nn.Linear(512, 256)
        self.bn7 = nn.BatchNorm1d(256)
        self.dp2 = nn.Dropout(p=args.dropout)
        self.linear3 = nn.Linear(256, output_channels)

    def forward(self, x):
        B = x.size(0)
        x1 = self.conv1(get_graph_feature(x, k=self.k)).max(dim=-1)[0]
        x2 = self.conv2(get_graph_feature(x1, k=self.k)).max(dim=-1)[0]
        x3 = self.conv3(get_graph_feature(x2, k=self.k)).max(dim=-1)[0]
        x4 = self.conv4(get_graph_feature(x3, k=self.k)).max(dim=-1)[0]
        x = self.conv5(torch.cat((x1, x2, x3, x4), dim=1))
        x = torch.cat((F.adaptive_max_pool1d(x, 1).view(B, -1), F.adaptive_avg_pool1d(x, 1).view(B, -1)), 1)
        x = self.dp1(F.leaky_relu(self.bn6(self.linear1(x)), negative_slope=0.2))
        x = self.dp2(F.leaky_relu(self.bn7(self.linear2(x)), negative_slope=0.2))
        x = F.log_softmax(self.linear3(x), -1)
        return x, x, x


# ----------------------------------------------------------------------------------------------------------
# GeoA3 (attack/GeoA3/{knn_utils,utility,loss_utils,GeoA3_attack}.py)
# ----------------------------------------------------------------------------------------------------------
class GeoA3Oracle:
    """Restatement of the GeoA3 loop. ``as_written=True`` reproduces knn_utils.py:12-15 literally (squared norms
    broadcast on the wrong axes) — used to pin this oracle against the real reference;
    ``as_written=False`` uses true squared distances — what the build implements and is compared against."""

    def __init__(self, as_written=True, dtype=None):
        self.as_written, self.dtype = as_written, dtype

    # -- knn_utils.py
    def knn_points(self, p1, p2, K=1):
        if self.dtype is not None:
            p1, p2 = p1.to(self.dtype), p2.to(self.dtype)
        inner = -2. * torch.matmul(p1, p2.transpose(2, 1))
        p1_2 = torch.sum((p1.transpose(2, 1)) ** 2, dim=1, keepdim=True)      # [B,1,N]
        p2_2 = torch.sum((p2.transpose(2, 1)) ** 2, dim=1, keepdim=True)      # [B,1,M]
        if self.as_written:
            dist = p1_2 + inner + p2_2.transpose(2, 1)                        # :15 (wrong axes; N must equal M)
        else:
            dist = p1_2.transpose(2, 1) + inner + p2_2
        value, pos = (-dist).topk(k=K, dim=-1)
        return (-value).float(), pos

    @staticmethod
    def knn_gather(x, idx):
        N, M, U = x.shape
        _, L, K = idx.shape
        return x[:, :, None].expand(-1, -1, K, -1).gather(1, idx[:, :, :, None].expand(-1, -1, -1, U))

    @staticmethod
    def _normalize(v, eps=1e-12):
        return v / v.norm(2, 1, keepdim=True).clamp(min=eps).expand_as(v)

    # -- utility.py:43-92
    def estimate_normal(self, pc, k):
        with torch.no_grad():
            b, _, n = pc.size()
            _, idx = self.knn_points(pc.permute(0, 2, 1), pc.permute(0, 2, 1), K=k + 1)
            nn_pts = self.knn_gather(pc.permute(0, 2, 1), idx).permute(0, 3, 1, 2)[:, :, :, 1:].contiguous()
            out = []
            for i in range(b):
                cps = nn_pts[i].detach().permute(1, 0, 2)
                cps = cps - torch.mean(cps, dim=2, keepdim=True)
                cov = (1.0 / (k - 1)) * torch.bmm(cps, cps.permute(0, 2, 1))
                ev, evec = torch.linalg.eigh(cov)
                nv = torch.gather(evec, 2, torch.argmin(ev, dim=1).unsqueeze(1).unsqueeze(2).expand(n, 3, 1)).squeeze()
                sign = -torch.sign(torch.bmm(nv.view(n, 1, 3), cps.sum(dim=2).view(n, 3, 1))).squeeze(2)
                out.append((sign * nv).permute(1, 0))
            return torch.stack(out, 0).float()

    # -- loss_utils.py
    def chamfer_loss(self, adv, ori):
        d1, _ = self.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), 1)
        d2, _ = self.knn_points(ori.permute(0, 2, 1), adv.permute(0, 2, 1), 1)
        return d1.squeeze(-1).mean(-1) + d2.squeeze(-1).mean(-1)

    def pseudo_chamfer_loss(self, adv, ori):
        return self.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), 1)[0].squeeze(-1).mean(-1)

    def hausdorff_loss(self, adv, ori):
        return self.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), 1)[0].squeeze(-1).max(-1)[0]

    def _kappa(self, pc, normal, k):
        _, idx = self.knn_points(pc.permute(0, 2, 1), pc.permute(0, 2, 1), k + 1)
        nn_pts = self.knn_gather(pc.permute(0, 2, 1), idx).permute(0, 3, 1, 2)[:, :, :, 1:].contiguous()
        vectors = self._normalize(nn_pts - pc.unsqueeze(3))
        return torch.abs((vectors * normal.unsqueeze(3)).sum(1)).mean(2)

    def kappa_adv(self, adv, ori, ori_normal, k):
        _, idx = self.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), 1)
        normal = self.knn_gather(ori_normal.permute(0, 2, 1), idx).permute(0, 3, 1, 2).squeeze(3).contiguous()
        return self._kappa(adv, normal, k), normal

    def curvature_loss(self, adv, ori, adv_kappa, ori_kappa):
        _, idx = self.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), 1)
        return ((adv_kappa - torch.gather(ori_kappa, 1, idx.squeeze(-1))) ** 2).mean(-1)

    # -- GeoA3_attack.py:103-183
    def forward_step(self, net, pc_ori, x, normal_ori, ori_kappa, target, scale_const, cfg, targeted):
        out = net(x)[0]
        if cfg.cls_loss_type == 'Margin':
            oh = torch.zeros(target.size() + (cfg.classes,)).scatter_(1, target.unsqueeze(1), 1.)
            fake = (oh * out).sum(1)
            other = ((1. - oh) * out - oh * 10000.).max(1)[0]
            cls_loss = torch.clamp((other - fake if targeted else fake - other) + cfg.confidence, min=0.)
        elif cfg.cls_loss_type == 'CE':
            ce = nn.CrossEntropyLoss(reduction='none')(out, target.long())
            cls_loss = ce if targeted else -ce
        else:
            cls_loss = torch.zeros(x.shape[0])
        if cfg.dis_loss_type == 'CD':
            dis = self.pseudo_chamfer_loss(x, pc_ori) if cfg.is_cd_single_side else self.chamfer_loss(x, pc_ori)
            constrain = cfg.dis_loss_weight * dis
        elif cfg.dis_loss_type == 'L2':
            dis = ((x - pc_ori) ** 2).sum(1).sum(1)
            constrain = cfg.dis_loss_weight * dis
        else:
            dis, constrain = 0, 0
        hd = 0
        if cfg.hd_loss_weight != 0:
            hd = self.hausdorff_loss(x, pc_ori)
            constrain = constrain + cfg.hd_loss_weight * hd
        curv = 0
        if cfg.curv_loss_weight != 0:
            ak, _ = self.kappa_adv(x, pc_ori, normal_ori, cfg.curv_loss_knn)
            curv = self.curvature_loss(x, pc_ori, ak, ori_kappa)
            constrain = constrain + cfg.curv_loss_weight * curv
        loss_n = cls_loss + scale_const.float() * constrain
        return out, loss_n.mean(), loss_n, cls_loss, dis, hd, curv, constrain

    # -- GeoA3_attack.py:185-473 (default mode: full offset variable, Adam, no jitter / projection / clip)
    def attack(self, net, pc, label, cfg, per_sample_label=False):
        targeted = cfg.attack_method != 'untarget'
        pc = pc.transpose(2, 1).float()
        normal = self.estimate_normal(pc, k=3)
        b, _, n = pc.size()
        pc_ori, normal_ori = pc.contiguous(), normal
        gt = label.view(-1)
        target = gt
        kappa_ori = self._kappa(pc_ori, normal_ori, cfg.curv_loss_knn) if cfg.curv_loss_weight != 0 else None
        lower, scale_const, upper = torch.zeros(b), torch.ones(b) * cfg.initial_const, torch.ones(b) * 1e10
        best_loss = [1e10] * b
        best_attack = torch.ones(b, 3, n)
        best_step = [-1] * b
        all_loss = [[-1] * b] * cfg.iter_max_steps
        labels_now = [0] * b
        for search_step in range(cfg.binary_max_steps):
            iter_best_loss, iter_best_score = [1e10] * b, [-1] * b
            constrain = torch.ones(b) * 1e10
            for step in range(cfg.iter_max_steps):
                if step == 0:
                    offset = torch.zeros(b, 3, n)
                    nn.init.normal_(offset, mean=0, std=1e-3)
                    offset.requires_grad_()
                    opt = torch.optim.Adam([offset], lr=cfg.lr)
                x = pc_ori + offset
                with torch.no_grad():
                    for k in range(b):
                        lab = int(torch.argmax(net(x[k].unsqueeze(0))[0]))
                        labels_now[k] = lab
                        ok = (lab == int(target[k])) if targeted else (lab != int(gt[k]))
                        metric = float(constrain[k])
                        if ok and metric < best_loss[k]:
                            best_loss[k], best_step[k] = metric, step
                            best_attack[k] = x.data[k].clone()
                        if ok and metric < iter_best_loss[k]:
                            iter_best_loss[k], iter_best_score[k] = metric, lab
                _, loss, loss_n, _, _, _, _, constrain = self.forward_step(net, pc_ori, x, normal_ori, kappa_ori, target,
                                                                          scale_const, cfg, targeted)
                all_loss[step] = loss_n.detach().tolist()
                opt.zero_grad()
                loss.backward()
                opt.step()
            for k in range(b):
                lab = labels_now[k] if per_sample_label else labels_now[-1]     # :395 uses the LAST sample's label (A-7)
                ok = (lab == int(target[k])) if targeted else (lab != int(gt[k]))
                if ok and iter_best_score[k] != -1:
                    lower[k] = max(lower[k], scale_const[k])
                    scale_const[k] = (lower[k] + upper[k]) * 0.5 if upper[k] < 1e9 else scale_const[k] * 2
                else:
                    upper[k] = min(upper[k], scale_const[k])
                    if upper[k] < 1e9:
                        scale_const[k] = (lower[k] + upper[k]) * 0.5
        return best_attack, target, (np.array(best_loss) < 1e10), best_step, all_loss


# ----------------------------------------------------------------------------------------------------------
# add-cluster / add-object functors (attack/CW/CW_utils/dist_utils.py:226-333) and GeoA3's
# uniform_loss (attack/GeoA3/loss_utils.py:159-197)
# ----------------------------------------------------------------------------------------------------------
def farthest_dist(adv_pc, weights=None, batch_avg=True):
    """dist_utils.py:226-254 — adv_pc [B, num_add, cl_num_p, 3]: farthest intra-cluster pair (+1e-7 on the
    differences, :244), summed over the clusters, times weights. Pinned by tests/golden/f4.npz."""
    B = adv_pc.shape[0]
    w = _w(weights, B, adv_pc)
    delta = adv_pc[:, :, None, :, :] - adv_pc[:, :, :, None, :] + 1e-7
    far = torch.norm(delta, p=2, dim=-1).max(dim=2)[0].max(dim=2)[0].sum(dim=1)
    loss = far * w
    return loss.mean() if batch_avg else loss


def far_chamfer_dist(adv_pc, ori_pc, num_add, method='adv2ori', chamfer_weight=0.1, weights=None, batch_avg=True, dtype=None):
    """:257-292 — FarthestDist of the clusters + chamfer_weight * ChamferDist(adv clusters, ori). Pinned by f4.npz."""
    B = adv_pc.shape[0]
    cd = ChamferDist(method=method, dtype=dtype)(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg)
    return farthest_dist(adv_pc.view(B, num_add, -1, 3), weights=weights, batch_avg=batch_avg) + cd * chamfer_weight


def l2_chamfer_dist(adv_pc, ori_pc, adv_obj, ori_obj, method='adv2ori', chamfer_weight=0.2, weights=None, batch_avg=True,
                    dtype=None):
    """:295-333 — L2Dist(adv objects, clean objects) + chamfer_weight * ChamferDist(placed objects, ori). Pinned by
    f4.npz."""
    B = adv_pc.shape[0]
    cd = ChamferDist(method=method, dtype=dtype)(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg)
    l2 = L2Dist()(adv_obj.reshape(B, -1, 3), ori_obj.reshape(B, -1, 3), weights=weights, batch_avg=batch_avg)
    return l2 + chamfer_weight * cd


def uniform_loss(adv_pc, percentages=(0.004, 0.006, 0.008, 0.010, 0.012), radius=1.0, k=2):
    """attack/GeoA3/loss_utils.py:159-197 — PARITY UNPINNED. The reference body calls furthest_point_sample /
    gather_operation / ball_query / grouping_operation on a module that defines none of them (SURVEY A-12), so it
    cannot run and no fixture of it exists; its default weight is 0 (Eval_GeoA3.py:166). This restates the INTENDED
    semantics those names have in the pointnet2 CUDA ops the code was written against: FPS from index 0; ball query =
    first `nsample` indices inside the radius, padded with the first; true squared distances for the in-group kNN.
    Line anchors: npoint :163, per-percentage constants :165-169, grouping :171-177, kNN + expected length :180-186,
    scaling :188-189, mean over percentages :196."""
    import math
    if adv_pc.size(1) == 3:
        adv_pc = adv_pc.permute(0, 2, 1).contiguous()
    b, n, _ = adv_pc.size()
    npoint = int(n * 0.05)
    orc = GeoA3Oracle(as_written=False, dtype=torch.float64)    # in-group distances in double: groups hold duplicates
    loss = None                                                 # (ball-query padding) whose true distance is 0
    for p in percentages:
        p = p * 4
        nsample = int(n * p)
        r = math.sqrt(p * radius)
        expect_len = math.sqrt(math.pi * (radius ** 2) * p / nsample)
        with torch.no_grad():
            fps_idx = farthest_point_sample(adv_pc, npoint, start=torch.zeros(b, dtype=torch.long))
            new_xyz = index_points(adv_pc, fps_idx)
            idx = query_ball_point(r, nsample, adv_pc, new_xyz, exact=True)
        grouped = index_points(adv_pc, idx).reshape(b * npoint, nsample, 3)
        d = orc.knn_points(grouped, grouped, K=k + 1)[0][:, :, 1:]
        d = torch.sqrt(torch.abs(d) + 1e-12).mean(dim=-1)
        d = ((d - expect_len) ** 2 / (expect_len + 1e-12)).reshape(-1)
        mean = d.mean() * math.pow(p * 100, 2)
        loss = mean if loss is None else loss + mean
    return loss / len(percentages)


# ----------------------------------------------------------------------------------------------------------
# AOF / TAOF (attack/AOF/TAOF_attack.py)
# ----------------------------------------------------------------------------------------------------------
def aof_knn(x, k):
    """TAOF_attack.py:13-28."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]


def get_Laplace_from_pc(ori_pc, k=30):
    """:31-52."""
    pc = ori_pc.detach().clone()
    with torch.no_grad():
        idx = aof_knn(pc, k)
        pc = pc.transpose(2, 1).contiguous()
        point_mat = pc.unsqueeze(2) - pc.unsqueeze(1)
        A = torch.exp(-torch.sum(point_mat.square(), dim=3))
        mask = torch.zeros_like(A)
        mask.scatter_(2, idx, 1)
        mask = mask + mask.transpose(2, 1)
        mask[mask > 1] = 1
        A = A * mask
        L = torch.diag_embed(torch.sum(A, dim=2)) - A
        e, v = torch.linalg.eigh(L)
    return e, v, L


def taof_attack(model, data, target, y_truth, adv_func, clip_func, attack_lr=1e-2, binary_step=2, num_iter=200,
                GAMMA=0.5, low_pass=100):
    """:83-244. Returns (o_bestdist [B] f64, adv [B,K,3] numpy, success_num)."""
    B, K = data.shape[:2]
    data = data.float().transpose(1, 2).contiguous()
    ori = data.clone().detach()
    target, y_truth = target.long().view(-1), y_truth.long().view(-1)
    label_val, y_val = target.numpy(), y_truth.numpy()
    o_bestdist = np.array([1e10] * B)
    o_bestscore = np.array([-1] * B)
    o_bestattack = np.zeros((B, 3, K))
    for p in model.parameters():
        p.requires_grad = False
    model(ori)
    input_val = None
    for bstep in range(binary_step):
        adv = ori.clone().detach() + torch.randn((B, 3, K)) * 1e-7
        _, V, _ = get_Laplace_from_pc(adv)
        projs = torch.bmm(adv, V)
        hfc = torch.bmm(projs[..., low_pass:], V[..., low_pass:].transpose(2, 1)).detach().clone()
        lfc = torch.bmm(projs[..., :low_pass], V[..., :low_pass].transpose(2, 1)).detach().clone()
        lfc.requires_grad_()
        opt = torch.optim.Adam([lfc], lr=attack_lr, weight_decay=0.)
        for it in range(num_iter):
            adv = lfc + hfc
            adv_loss = (1 - GAMMA) * adv_func(model(adv)[0], target).mean()
            opt.zero_grad()
            adv_loss.backward()
            (GAMMA * adv_func(model(lfc)[0], target).mean()).backward()
            opt.step()
            with torch.no_grad():
                adv = lfc + hfc
                adv.data = clip_func(adv.detach().clone(), ori)
                coeff = torch.bmm(adv, V)
                hfc.data = torch.bmm(coeff[..., low_pass:], V[..., low_pass:].transpose(2, 1))
                lfc.data = torch.bmm(coeff[..., :low_pass], V[..., :low_pass].transpose(2, 1))
                pred = torch.argmax(model(adv)[0], dim=1).numpy()
                lfc_pred = torch.argmax(model(lfc)[0], dim=1).numpy()
            dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).detach().numpy()
            input_val = adv.detach().numpy().copy()
            for e in range(B):
                if dist_val[e] < o_bestdist[e] and pred[e] == label_val[e] and lfc_pred[e] != y_val[e]:
                    o_bestdist[e], o_bestscore[e] = dist_val[e], pred[e]
                    o_bestattack[e] = input_val[e]
    fail = o_bestscore < 0
    o_bestattack[fail] = input_val[fail]
    adv_pc = torch.tensor(o_bestattack).float()
    preds = torch.argmax(model(adv_pc)[0], dim=-1)
    return o_bestdist, adv_pc.numpy().transpose((0, 2, 1)), int((preds == target).sum())


# ---------------------------------------------------------------------------------------------------------
# attack/additional_exp/CW_attack.py: CW with the binary-searched point-set distance, z-only
# perturbation in a +-0.4 box, renormalisation in the loop, expectation over 10 small rotations (+ resampling).
# Restated for B = 1 semantics generalised per sample (the reference's prints/zip force B = 1, :80-84,:158-182).
# ---------------------------------------------------------------------------------------------------------
def _renorm_cf(x):
    """attack/additional_exp/CW_attack.py:107-115."""
    p = x.permute(0, 2, 1)
    p = p - torch.mean(p, dim=1).unsqueeze(1)
    var = torch.max(torch.sqrt(torch.sum(p ** 2, dim=2)), dim=1, keepdim=True)[0]
    return (p / var.unsqueeze(1)).permute(0, 2, 1)


def cw_additional_attack(model, data, target, origin_label, adv_func, dist_func, attack_lr=1e-2, init_weight=10.,
                         max_weight=80., binary_step=10, num_iter=500, whether_target=True, whether_1d=True,
                         whether_renormalization=False, whether_3Dtransform=False, whether_resample=False):
    """adv_func(logits, label, whether_target=...), dist_func(adv[B,K,3], ori[B,K,3], weights[B]) -> scalar or [B].
    Returns (o_bestdist [B] f64, o_bestattack [B,K,3] f64, success_num). Random streams as the reference: torch CPU
    generator for the start noise and the angles (:88,:196), Python `random` for the axis choice / resampling (:211,:238)."""
    import random
    if data.shape[2] == 3:
        data = data.transpose(1, 2).contiguous()
    B, _, K = data.shape
    data = data.float().detach()
    ori = data.clone().detach()
    target = target.long().view(-1)
    origin_label = origin_label.long().view(-1)
    goal = (target if whether_target else origin_label).numpy()
    goal_t = target if whether_target else origin_label
    lower, upper, cur_w = np.zeros((B,)), np.ones((B,)) * max_weight, np.ones((B,)) * init_weight
    o_bestdist = np.array([1e10] * B)
    o_bestscore = np.array([-1] * B)
    o_bestattack = np.zeros((B, 3, K))
    model(ori)                                                                    # :75 clean forward
    adv = ori.clone().detach() + torch.randn((B, 3, K)) * 1e-7                    # :88 — ONE start for the whole search
    input_val = None
    box = 0.4
    for bstep in range(binary_step):
        adv.requires_grad_()
        bestdist = np.array([1e10] * B)
        bestscore = np.array([-1] * B)
        opt = torch.optim.Adam([adv], lr=attack_lr, weight_decay=0.)              # :96
        for it in range(num_iter):
            logits = model(_renorm_cf(adv) if whether_renormalization else adv)[0]   # :105-117
            pred = torch.argmax(logits, dim=1)
            dist_loss = dist_func(adv.permute(0, 2, 1), ori.permute(0, 2, 1), torch.from_numpy(cur_w))   # :151-153
            dist_val = np.broadcast_to(dist_loss.detach().double().numpy().reshape(-1), (B,))
            pred_val = pred.detach().numpy()
            input_val = adv.detach().numpy().copy()
            dist_loss = dist_loss.mean()
            for e in range(B):                                                    # :160-182
                ok = (pred_val[e] == goal[e]) if whether_target else (pred_val[e] != goal[e])
                if dist_val[e] < bestdist[e] and ok:
                    bestdist[e], bestscore[e] = dist_val[e], pred_val[e]
                if dist_val[e] < o_bestdist[e] and ok:
                    o_bestdist[e], o_bestscore[e] = dist_val[e], pred_val[e]
                    o_bestattack[e] = input_val[e]
            wt = 1 if whether_target else 0
            if whether_3Dtransform:                                               # :191-251
                diff = adv - ori.detach()
                losses = []
                for _ in range(10):
                    theta = torch.randn(1) * 1e-2
                    c, s_ = float(torch.cos(theta)), float(torch.sin(theta))
                    r = random.random()
                    if r < 0.2:
                        m = [[c, s_, 0], [-s_, c, 0], [0, 0, 1]]
                    elif r < 0.4:
                        m = [[1, 0, 0], [0, c, s_], [0, -s_, c]]
                    elif r < 0.6:
                        m = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]
                    else:
                        m = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
                    Tr = torch.tensor(m, dtype=torch.float32).unsqueeze(0).expand(B, -1, -1)
                    x = torch.bmm(Tr, ori.detach()) + diff
                    if whether_renormalization:
                        x = _renorm_cf(x)
                    if whether_resample:
                        idx = random.sample(range(1, K * 2), K)                   # :238 (4000 of 8000 there)
                        x = torch.index_select(torch.cat((x, x), 2), 2, torch.LongTensor(idx))
                    losses.append(adv_func(model(x)[0], goal_t, whether_target=wt).mean())
                adv_loss = torch.mean(torch.stack(losses))
            else:
                adv_loss = adv_func(logits, goal_t, whether_target=wt).mean()     # :254-260
            loss = adv_loss + dist_loss
            opt.zero_grad()
            loss.backward()
            opt.step()
            if whether_1d:                                                        # :266-275
                with torch.no_grad():
                    adv[:, 0] = ori[:, 0]
                    adv[:, 1] = ori[:, 1]
                    adv[:, 2] = torch.max(torch.min(adv[:, 2], ori[:, 2] + box), ori[:, 2] - box)
        for e in range(B):                                                        # :283-303
            hit = (bestscore[e] == goal[e]) if whether_target else (bestscore[e] != goal[e])
            if hit and bestscore[e] != -1 and bestdist[e] <= o_bestdist[e]:
                lower[e] = max(lower[e], cur_w[e])
            else:
                upper[e] = min(upper[e], cur_w[e])
            cur_w[e] = (lower[e] + upper[e]) / 2.
    fail_idx = (lower == 0.)                                                      # :309-310
    o_bestattack[fail_idx] = input_val[fail_idx]
    return o_bestdist, o_bestattack.transpose((0, 2, 1)), int((lower > 0.).sum())
