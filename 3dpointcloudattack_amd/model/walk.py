"""Guided curve walk of CurveNet — MI355X mirror of model/walk.py (``Walk``; same parameters / state_dict keys).

Same algorithm as the reference (model/walk.py:74-153): ``curve_length`` steps; at every step each of the ``curve_num``
curves scores the k neighbours of its current node with a 1x1 conv on [neighbour feature ; curve descriptor], damps
directions that would fold back (crossover suppression), picks the arg-max neighbour with a straight-through hard
softmax and blends its descriptor with a learned momentum. Written with batched gathers on the device the tensors
live on (the reference hard-codes torch.device('cuda') and flattened batch offsets, :84-90).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def pw(mod, x):
    from .curvenet_util import pw as _pw   # late import: curvenet_util imports this module
    return _pw(mod, x)


def batched_index_select(input, dim, index):
    """model/walk.py:7-14."""
    views = [input.shape[0]] + [1 if i != dim else -1 for i in range(1, len(input.shape))]
    expanse = list(input.shape)
    expanse[0] = -1
    expanse[dim] = -1
    return torch.gather(input, dim, index.view(views).expand(expanse))


def gumbel_softmax(logits, dim, temperature=1):
    """model/walk.py:17-33 — straight-through hard softmax (no Gumbel noise): one-hot forward, softmax backward."""
    y = F.softmax(logits / temperature, dim=dim)
    hard = torch.zeros_like(y).scatter_(-1, y.argmax(dim=-1, keepdim=True), 1.0)
    return (hard - y).detach() + y


class Walk(nn.Module):
    """Walk in the cloud (model/walk.py:35-153)."""

    def __init__(self, in_channel, k, curve_num, curve_length):
        super(Walk, self).__init__()
        self.curve_num = curve_num
        self.curve_length = curve_length
        self.k = k
        self.agent_mlp = nn.Sequential(nn.Conv2d(in_channel * 2, 1, kernel_size=1, bias=False), nn.BatchNorm2d(1))
        self.momentum_mlp = nn.Sequential(nn.Conv1d(in_channel * 2, 2, kernel_size=1, bias=False), nn.BatchNorm1d(2))
        self.fused = True

    def crossover_suppression(self, cur, neighbor, bn, n, k):
        """:55-72 — 1 + cos(angle between the last move and each candidate move), clamped to [0,1]; no gradient."""
        neighbor = neighbor.detach()
        cur = cur.unsqueeze(-1).detach()                       # [M,C,1]
        dot = torch.bmm(cur.transpose(1, 2), neighbor)         # [M,1,k]
        divider = torch.clamp(torch.norm(cur, dim=1, keepdim=True) * torch.norm(neighbor, dim=1, keepdim=True), min=1e-8)
        ans = torch.div(dot, divider).squeeze(1)               # [M,k]
        return torch.clamp(1. + ans, 0., 1.0).detach()

    def forward(self, xyz, x, adj, cur, cl=False):
        """Reference layout (default): x [B,C,N] features, adj [B,N,k] neighbour indices (self excluded), cur
        [B,curve_num,1] start nodes -> curves [B,C,curve_num,curve_length]. cl=True: x [B,N,C], adj int32 [B,N,k], cur
        int32 [B,curve_num] -> [B,curve_num,curve_length,C]. One HIP launch per direction (pc3d_curve_walk_*_f32)
        when the channel count is one the kernel is built for; the step-by-step formulation below otherwise."""
        C = x.shape[2] if cl else x.shape[1]
        B = x.shape[0]
        if self.fused and x.is_cuda and C in ops.CURVE_WALK_CHANNELS and adj.shape[2] <= 64:
            from .curvenet_util import folded_pw
            _, _, aw, ab = folded_pw(self.agent_mlp)
            _, _, mw, mb = folded_pw(self.momentum_mlp)
            feats = x.float().contiguous() if cl else x.transpose(1, 2).contiguous().float()
            curves = ops.curve_walk(feats, adj.to(torch.int32), cur.reshape(B, self.curve_num).to(torch.int32), aw, ab,
                                    mw, mb, self.curve_length)
            return curves if cl else curves.permute(0, 3, 1, 2)    # [B,cn,L,C] (-> [B,C,cn,L])
        if cl:
            cf = self.forward_steps(xyz, x.transpose(1, 2), adj.long(), cur.long().reshape(B, self.curve_num, 1))
            return cf.permute(0, 2, 3, 1).contiguous()
        return self.forward_steps(xyz, x, adj, cur)

    def forward_steps(self, xyz, x, adj, cur):
        """The same walk as ~30 torch launches per step (model/walk.py:74-153 restated with batched gathers)."""
        B, C, N = x.size()
        cn, k = self.curve_num, self.k
        feats = x.transpose(1, 2).contiguous()                 # [B,N,C]
        node = cur.reshape(B, cn)                              # current node of every curve
        adj = adj.long()
        curves = []
        pre_feature = cur_feature = None
        cur_flat = pre_flat = None
        for step in range(self.curve_length):
            if step == 0:
                start = torch.gather(feats, 1, node.unsqueeze(-1).expand(-1, -1, C))      # [B,cn,C]
                pre_feature = start.transpose(1, 2).unsqueeze(-1)                         # [B,C,cn,1]
            else:
                both = torch.cat((cur_feature.squeeze(3), pre_feature.squeeze(3)), dim=1)  # [B,2C,cn]
                mom = F.softmax(pw(self.momentum_mlp, both), dim=1).view(B, 1, cn, 2)          # dynamic momentum
                pre_feature = torch.sum(torch.cat((cur_feature, pre_feature), dim=-1) * mom, dim=-1, keepdim=True)
                pre_flat = pre_feature.transpose(1, 2).contiguous().view(B * cn, C)
            nbr_idx = torch.gather(adj, 1, node.unsqueeze(-1).expand(-1, -1, k))           # [B,cn,k]
            nbr = torch.gather(feats, 1, nbr_idx.reshape(B, cn * k, 1).expand(-1, -1, C)).view(B, cn, k, C)
            nbr_flat = nbr.reshape(B * cn, k, C).transpose(1, 2).contiguous()              # [B*cn,C,k]
            nbr_c = nbr.permute(0, 3, 1, 2)                                                # [B,C,cn,k]
            score = pw(self.agent_mlp, torch.cat((nbr_c, pre_feature.expand_as(nbr_c)), dim=1))  # [B,1,cn,k]
            if step != 0:
                d = self.crossover_suppression(cur_flat - pre_flat, nbr_flat - cur_flat.unsqueeze(-1), B, cn, k)
                score = torch.mul(score, d.view(B, cn, k).unsqueeze(1))
            pick = gumbel_softmax(score, -1)                                               # [B,1,cn,k] one-hot (ST)
            cur_feature = torch.sum(nbr_c * pick, dim=-1, keepdim=True)                    # [B,C,cn,1]
            cur_flat = cur_feature.transpose(1, 2).contiguous().view(B * cn, C)
            choice = torch.argmax(pick, dim=-1).view(B, cn, 1)
            node = torch.gather(nbr_idx, 2, choice).squeeze(2)
            curves.append(cur_feature)
        return torch.cat(curves, dim=-1)
