"""CurveNet building blocks — MI355X mirror of model/curvenet_util.py (same class names, constructor arguments,
sub-module names / ``state_dict`` keys). Neighbour search, farthest-point sampling (start index 0, :81), ball query and
every gather run on the HIP kernels (pc3d_knn_f32 / pc3d_fps_f32 / pc3d_ball_query_f32 / pc3d_group_gather_f32); the
reference materialises [B,N,N] distance matrices for each of them (nine kNNs per forward) and hard-codes
torch.device('cuda').
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .pointnet import _FrozenFusedMixin
from .walk import Walk


def _fold_pointwise(conv, bn):
    """(W [O,C], b [O] or None) of a kernel-size-1 Conv1d/Conv2d or a Linear followed by an eval-mode BatchNorm
    (bn may be None), with DETACHED parameters: the attacks differentiate with respect to the points only."""
    w = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
    b = conv.bias.detach().float() if conv.bias is not None else None
    if bn is not None:
        sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * sc[:, None]
        b = (b if b is not None else 0.0) * sc + (bn.bias.detach().float() - bn.running_mean.detach().float() * sc)
    return w.contiguous(), (b.contiguous() if b is not None else None)


def folded_pw(mod, bn=None):
    """(conv, activation, W, b) of a frozen pointwise layer: `mod` is a 1x1 Conv1d / Conv2d / Linear or an
    nn.Sequential(conv[, bn][, activation]). Folded weights are cached on the module and re-folded whenever a
    parameter / buffer changes."""
    act = None
    conv = mod
    if isinstance(mod, nn.Sequential):
        conv = mod[0]
        for m in list(mod)[1:]:
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                bn = m
            else:
                act = m
    if bn is not None and bn.training:
        raise NotImplementedError("CurveNet: only the eval-mode (frozen-weight) attack path is implemented on MI355X")
    ts = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
    key = tuple((t.data_ptr(), t._version) for t in ts if t is not None)
    cache = getattr(conv, "_pw_cache", None)
    if cache is None or cache[0] != key:
        cache = (key,) + _fold_pointwise(conv, bn)
        object.__setattr__(conv, "_pw_cache", cache)
    return conv, act, cache[1], cache[2]


def pw(mod, x, bn=None):
    """Frozen pointwise layer: `mod` as in folded_pw; x is [B,C,L], [B,C,H,W] (conv) or [B,C] (Linear). One GEMM
    against folded weights instead of MIOpen convolution + batch-norm + weight-gradient kernels (the reference leaves
    requires_grad on and pays for dL/dW in every attack step)."""
    _, act, w, b = folded_pw(mod, bn)
    if x.dim() == 2:
        y = ops.linear_act(x, w, b)
    else:
        shp = x.shape
        y = torch.matmul(w, x.reshape(shp[0], shp[1], -1))
        if b is not None:
            y = y + b[None, :, None]
        y = y.view(shp[0], w.shape[0], *shp[2:])
    if act is None:
        return y
    if isinstance(act, nn.LeakyReLU):
        return F.leaky_relu(y, act.negative_slope)
    if isinstance(act, nn.ReLU):
        return F.relu(y)
    if isinstance(act, nn.Sigmoid):
        return torch.sigmoid(y)
    raise NotImplementedError(f"pw: unsupported activation {type(act).__name__}")


def _act_of(act):
    """(name, slope) of a folded layer's activation module for ops.linear_act."""
    if act is None:
        return None, 0.0
    if isinstance(act, nn.LeakyReLU):
        return "leaky", act.negative_slope
    if isinstance(act, nn.ReLU):
        return "relu", 0.0
    raise NotImplementedError(f"pw_cl: unsupported activation {type(act).__name__}")


def pw_cl(mod, x, bn=None, res=None, act=None):
    """Frozen pointwise layer on channels-last rows x [..., C] -> [..., O] (one fp32-MFMA launch; `mod` as in
    folded_pw). res [..., O]: a residual branch summed in before the activation; act = (name, slope) overrides the
    activation found in `mod` (the block's trailing LeakyReLU after a residual sum)."""
    _, a, w, b = folded_pw(mod, bn)
    name, slope = act if act is not None else _act_of(a)
    if res is not None:
        return ops.linear_res_act(x, w, b, res, name, slope)
    return ops.linear_act(x, w, b, name, slope)


def _cf(x):
    """[B,N,C] -> contiguous [B,C,N]."""
    return x.transpose(2, 1).contiguous()


def _cl(x):
    """[B,C,N] -> contiguous [B,N,C]."""
    return x.transpose(2, 1).contiguous().float()


def knn(x, k):
    """curvenet_util.py:10-17 — k+1 nearest (self first) of xyz [B,3,N] -> [B,N,k+1] int64."""
    xt = _cl(x)
    return ops.knn_raw(xt, xt, k + 1)[1].long()


def normal_knn(x, k):
    """:20-26."""
    xt = _cl(x)
    return ops.knn_raw(xt, xt, k)[1].long()


def pc_normalize(pc):
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def square_distance(src, dst):
    """:38-47 (dense; direct-difference form)."""
    return ops.pairwise(src.float(), dst.float())


def index_points(points, idx):
    """:50-66."""
    idx32 = idx.to(torch.int32)
    if idx.dim() == 2:
        B, S = idx.shape
        return ops.group_gather(None, points.float(), idx32.reshape(B, S, 1).contiguous()).view(B, S, -1)
    return ops.group_gather(None, points.float(), idx32.contiguous())


def hold_rng_position(B, N):
    """The reference's CurveNet FPS starts at ``torch.randint(0, N, (B,)) * 0`` (:81): index 0, but torch's global CPU
    generator still advances on every call. The same draw is made (and discarded) here so that seeded runs consume
    the identical stream — it feeds CW's start noise and PointNet++'s FPS starts later on. CPU-only, no device work."""
    torch.randint(0, N, (B,), dtype=torch.long)


def farthest_point_sample(xyz, npoint):
    """:69-90 — deterministic: starts at index 0 (:81)."""
    hold_rng_position(xyz.shape[0], xyz.shape[1])
    return ops.fps(xyz.float(), npoint, None).long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    """:93-113."""
    return ops.ball_query(radius, nsample, xyz.float(), new_xyz.float()).long()


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """:116-140 — new_xyz [B,npoint,3], new_points [B,npoint,nsample,D] (features only, not centred)."""
    xyz = xyz.float()
    hold_rng_position(xyz.shape[0], xyz.shape[1])
    fps_idx = ops.fps(xyz, npoint, None)
    B = xyz.shape[0]
    new_xyz = ops.group_gather(xyz, None, fps_idx.view(B, npoint, 1)).view(B, npoint, 3)
    idx = ops.ball_query(radius, nsample, xyz, new_xyz)
    new_points = ops.group_gather(None, points.float(), idx)
    if returnfps:
        return new_xyz, new_points, idx.long()
    return new_xyz, new_points


class Attention_block(nn.Module):
    """:143-170 (attention U-Net gate; segmentation only)."""

    def __init__(self, F_g, F_l, F_int):
        super(Attention_block, self).__init__()
        self.W_g = nn.Sequential(nn.Conv1d(F_g, F_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm1d(F_int))
        self.W_x = nn.Sequential(nn.Conv1d(F_l, F_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm1d(F_int))
        self.psi = nn.Sequential(nn.Conv1d(F_int, 1, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm1d(1),
                                 nn.Sigmoid())

    def forward(self, g, x):
        psi = pw(self.psi, F.leaky_relu(pw(self.W_g, g) + pw(self.W_x, x), negative_slope=0.2))
        return psi, 1. - psi


class LPFA(nn.Module):
    """Local point-feature aggregation (:175-236)."""

    def __init__(self, in_channel, out_channel, k, mlp_num=2, initial=False):
        super(LPFA, self).__init__()
        self.k = k
        self.initial = initial
        if not initial:
            self.xyz2feature = nn.Sequential(nn.Conv2d(9, in_channel, kernel_size=1, bias=False), nn.BatchNorm2d(in_channel))
        layers = []
        for _ in range(mlp_num):
            layers.append(nn.Sequential(nn.Conv2d(in_channel, out_channel, 1, bias=False), nn.BatchNorm2d(out_channel),
                                        nn.LeakyReLU(0.2)))
            in_channel = out_channel
        self.mlp = nn.Sequential(*layers)

    def _kernel_form(self, C):
        if self.initial:
            return len(self.mlp) == 1 and self.mlp[0][0].out_channels % 4 == 0
        return C % 4 == 0 and all(l[0].out_channels % 4 == 0 for l in self.mlp)

    def forward(self, x, xyz, idx=None, cl=False):
        """Reference layout (default): x [B,C,N], xyz [B,3,N], idx [B,N,>=k] or None -> [B,C',N]. cl=True: channels-last
        x [B,N,C] (ignored by the initial block), xyz [B,N,3], idx int32 [B,N,k] -> [B,N,C'] (what CurveNet.forward uses:
        no layout changes between the kernels)."""
        if cl:
            return self._forward_cl(x, xyz, idx)
        if xyz.is_cuda and self._kernel_form(0 if x is None else x.shape[1]):
            idx32 = None if idx is None else idx[:, :, :self.k].to(torch.int32).contiguous()
            return _cf(self._forward_cl(None if self.initial else _cl(x), _cl(xyz), idx32))
        x = self.group_feature(x, xyz, idx)
        for layer in self.mlp:
            x = pw(layer, x)
        return x.max(dim=-1, keepdim=False)[0] if self.initial else x.mean(dim=-1, keepdim=False)

    def _forward_cl(self, x, pts, idx32):
        if idx32 is None:
            idx32 = ops.knn_raw(pts.detach(), pts.detach(), self.k)[1]          # the k nearest, self first
            if self.initial:
                object.__setattr__(self, "_last_idx", idx32)                    # (CurveNet.forward publishes it)
        elif self.initial:
            object.__setattr__(self, "_last_idx", None)
        if not self._kernel_form(0 if x is None else x.shape[2]):
            return _cl(self.forward(None if x is None else _cf(x), _cf(pts), idx32.long()))
        return self._initial_edge_max(pts, idx32) if self.initial else self._edge_act_mean(x, pts, idx32)

    def _derived(self, name, src, build):
        """Tensors derived from folded weights, rebuilt when the folded tensor object changes."""
        hit = self.__dict__.get(name)
        if hit is None or hit[0] is not src:
            hit = (src,) + tuple(build())
            object.__setattr__(self, name, hit)
        return hit[1:]

    def _initial_edge_max(self, pts, idx32):
        """The first LPFA (:199-203, :226-236 with initial=True and one MLP layer) without the [B,32,N,k] edge tensor
        (335 MB at B=32, N=4096): W [p_i ; p_j ; p_j - p_i] = (Wb + Wc) p_j + (Wa - Wc) p_i, and LeakyReLU is monotone,
        so max_j leaky(bn(W e_ij)) = leaky(max_j P_j + Q_i): one [N,3] x [3,2C] GEMM + pc3d_edge_max_f32."""
        _, _, w, b = folded_pw(self.mlp[0])
        wpq, bpq = self._derived("_pq_w", w, lambda: (torch.cat((w[:, 3:6] + w[:, 6:9], w[:, 0:3] - w[:, 6:9]), 0).contiguous(),
                                                       torch.cat((torch.zeros_like(b), b))))
        PQ = ops.linear_act(pts, wpq, bpq)
        return ops.edge_max(PQ, idx32, self.mlp[0][2].negative_slope)           # [B,N,C]

    def _edge_act_mean(self, x, pts, idx32):
        """:204-236 with initial=False, channels-last and without the geometry tensor: xyz2feature is linear in
        [p_i ; p_j ; p_j - p_i], so  (x_j - x_i) + G geo_ij + t = A_j + B_i  with A = x + (Gb + Gc) p and
        B = (Ga - Gc) p + t - x (pc3d_lpfa_prep_f32).  One launch builds leaky(A_j + B_i) [B,N,k,C] (pc3d_edge_act_f32),
        the MLP is a channels-last GEMM per layer and the last LeakyReLU is fused with the neighbour mean
        (pc3d_act_mean_f32)."""
        _, _, g, t = folded_pw(self.xyz2feature)
        g1, g2 = self._derived("_geo_w", g, lambda: ((g[:, 3:6] + g[:, 6:9]).contiguous(), (g[:, 0:3] - g[:, 6:9]).contiguous()))
        A, Bc = ops.lpfa_prep(x, pts, g1, g2, t)
        if len(self.mlp) == 1:
            _, act, w, b = folded_pw(self.mlp[0])
            if b is not None and ops.lpfa_fused_supported(A.shape[2], w.shape[0], idx32.shape[2]):
                # one layer of equal width (every CIC block of the classifier): gather, activation, 1x1 conv,
                # activation and neighbour mean in ONE launch each way, no [B,N,k,C] tensor
                return ops.lpfa_fused(A, Bc, idx32, w, b, 0.2, act.negative_slope)
        E = ops.edge_act(A, Bc, idx32, 0.2)                                    # [B,N,k,C]
        for li, layer in enumerate(self.mlp):
            _, act, w, b = folded_pw(layer)
            last = li + 1 == len(self.mlp)       # the last layer's LeakyReLU is fused with the neighbour mean below
            E = ops.linear_act(E, w, b, None if last else "leaky", 0.0 if last else act.negative_slope)
        return ops.act_mean(E, self.mlp[-1][2].negative_slope)                  # [B,N,C']

    def group_feature(self, x, xyz, idx):
        """[B,9,N,k] = [p_i, p_j, p_j - p_i] (initial) or leaky((x_j - x_i) + xyz2feature(that)) [B,C,N,k]."""
        B, C, N = (xyz if x is None else x).size()
        if idx is None:
            idx = knn(xyz, k=self.k)[:, :, :self.k]
        idx32 = idx[:, :, :self.k].to(torch.int32).contiguous()
        pts = _cl(xyz)                                             # [B,N,3]
        nbr = ops.group_gather(pts, None, idx32)                   # [B,N,k,3] (differentiable in the points)
        ctr = pts.view(B, N, 1, 3).expand(-1, -1, self.k, -1)
        geo = torch.cat((ctr, nbr, nbr - ctr), dim=3).permute(0, 3, 1, 2).contiguous()   # [B,9,N,k]
        if self.initial:
            return geo
        feats = _cl(x)                                             # [B,N,C]
        rel = ops.group_gather(None, feats, idx32) - feats.view(B, N, 1, C)              # x_j - x_i
        return F.leaky_relu(rel.permute(0, 3, 1, 2) + pw(self.xyz2feature, geo), 0.2)


class PointNetFeaturePropagation(nn.Module):
    """:239-299 (segmentation decoder; not used by the classifiers / attacks)."""

    def __init__(self, in_channel, mlp, att=None):
        super(PointNetFeaturePropagation, self).__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        self.att = None
        if att is not None:
            self.att = Attention_block(F_g=att[0], F_l=att[1], F_int=att[2])
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel

    def forward(self, xyz1, xyz2, points1, points2):
        x1, x2, p2 = _cl(xyz1), _cl(xyz2), _cl(points2)
        B, N, _ = x1.shape
        if x2.shape[1] == 1:
            interpolated = p2.repeat(1, N, 1)
        else:
            dists, idx = ops.knn(x1, x2, 3)
            recip = 1.0 / (dists + 1e-8)
            weight = recip / torch.sum(recip, dim=2, keepdim=True)
            interpolated = torch.sum(ops.group_gather(None, p2, idx) * weight.view(B, N, 3, 1), dim=2)
        if self.att is not None:
            psix, _ = self.att(interpolated.permute(0, 2, 1), points1)
            points1 = points1 * psix
        new_points = interpolated if points1 is None else torch.cat([points1.permute(0, 2, 1), interpolated], dim=-1)
        new_points = new_points.permute(0, 2, 1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_points = F.leaky_relu(pw(conv, new_points, bn=bn), 0.2)
        return new_points


class CIC(nn.Module):
    """Curve intervention convolution block (:302-376)."""

    def __init__(self, npoint, radius, k, in_channels, output_channels, bottleneck_ratio=2, mlp_num=2, curve_config=None):
        super(CIC, self).__init__()
        self.in_channels = in_channels
        self.output_channels = output_channels
        self.bottleneck_ratio = bottleneck_ratio
        self.radius = radius
        self.k = k
        self.npoint = npoint
        planes = in_channels // bottleneck_ratio
        self.use_curve = curve_config is not None
        if self.use_curve:
            self.curveaggregation = CurveAggregation(planes)
            self.curvegrouping = CurveGrouping(planes, k, curve_config[0], curve_config[1])
        self.conv1 = nn.Sequential(nn.Conv1d(in_channels, planes, kernel_size=1, bias=False),
                                   nn.BatchNorm1d(in_channels // bottleneck_ratio),
                                   nn.LeakyReLU(negative_slope=0.2, inplace=True))
        self.conv2 = nn.Sequential(nn.Conv1d(planes, output_channels, kernel_size=1, bias=False),
                                   nn.BatchNorm1d(output_channels))
        if in_channels != output_channels:
            self.shortcut = nn.Sequential(nn.Conv1d(in_channels, output_channels, kernel_size=1, bias=False),
                                          nn.BatchNorm1d(output_channels))
        self.relu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        self.maxpool = MaskedMaxPool(npoint, radius, k)
        self.lpfa = LPFA(planes, planes, k, mlp_num=mlp_num, initial=False)

    def forward(self, xyz, x, cl=False, geo=None):
        """Reference layout (default): xyz [B,3,N], x [B,C,N] -> (xyz' [B,3,N'], [B,C',N']). cl=True: channels-last
        xyz [B,N,3], x [B,N,C] -> ([B,N',3], [B,N',C']); the block itself always runs channels-last. geo: this block's
        entry of CurveNet._geometry (sampling / ball-query / graph indices computed ahead on another stream)."""
        if cl:
            return self._forward_cl(xyz, x, geo)
        pts, y = self._forward_cl(_cl(xyz), _cl(x))
        return (xyz if pts.shape[1] == xyz.shape[2] else _cf(pts)), _cf(y)

    def _graph(self, pts):
        """(idx [B,N,k+1] self first, the same without the self column, its first k columns), int32. Consecutive
        blocks at one resolution receive the SAME point tensor object (it is returned unchanged below), so within one
        CurveNet.forward (which hands every block a fresh dict) the graph is computed once per resolution."""
        cache = self.__dict__.get("_graph_cache")
        hit = cache.get((id(pts), self.k)) if cache is not None else None
        if hit is not None and hit[0] is pts:
            return hit[1]
        pd = pts.detach()
        g = ops.knn_graph(pd, self.k)                   # (idx, idx[:, :, 1:], idx[:, :, :k]) from one launch
        if torch.is_grad_enabled() and pts.requires_grad:
            ops.attach_rev_index(g[2], pts.shape[1])     # shared by the LPFA blocks of this resolution (deterministic backward)
        if cache is not None:
            cache[(id(pts), self.k)] = (pts, g)
        return g

    def _forward_cl(self, pts, x, geo=None):
        pool = graph = ev_graph = None
        if geo is not None:
            pool, graph, ev_pool, ev_graph = geo
            if ev_pool is not None:
                torch.cuda.current_stream().wait_event(ev_pool)
        if pts.shape[1] != self.npoint:                            # FPS + ball-query max-pool down-sampling
            pts, x = self.maxpool(pts, x, cl=True, geo=pool)
        shortcut = x
        x = pw_cl(self.conv1, x)
        if ev_graph is not None:                                   # the kNN graph is built while the layers above run
            torch.cuda.current_stream().wait_event(ev_graph)
        _, adj, nbr = graph if graph is not None else self._graph(pts)
        if self.use_curve:
            curves = self.curvegrouping(x, pts, adj, cl=True)      # adj: no self-loops
            x = self.curveaggregation(x, curves, cl=True)
        x = self.lpfa(x, pts, nbr, cl=True)
        if self.in_channels != self.output_channels:
            shortcut = pw_cl(self.shortcut, shortcut)
        # relu(conv2(x) + shortcut) in conv2's epilogue
        return pts, pw_cl(self.conv2, x, res=shortcut, act=("leaky", self.relu.negative_slope))


class CurveAggregation(nn.Module):
    """:379-437 — fuse curve features back into the point features by inter- / intra-curve attention."""

    def __init__(self, in_channel):
        super(CurveAggregation, self).__init__()
        self.in_channel = in_channel
        mid_feature = in_channel // 2
        self.conva = nn.Conv1d(in_channel, mid_feature, kernel_size=1, bias=False)
        self.convb = nn.Conv1d(in_channel, mid_feature, kernel_size=1, bias=False)
        self.convc = nn.Conv1d(in_channel, mid_feature, kernel_size=1, bias=False)
        self.convn = nn.Conv1d(mid_feature, mid_feature, kernel_size=1, bias=False)
        self.convl = nn.Conv1d(mid_feature, mid_feature, kernel_size=1, bias=False)
        self.convd = nn.Sequential(nn.Conv1d(mid_feature * 2, in_channel, kernel_size=1, bias=False), nn.BatchNorm1d(in_channel))
        self.line_conv_att = nn.Conv2d(in_channel, 1, kernel_size=1, bias=False)
        self.fused = True

    def _kernel_form(self, x, cn, cl, C):
        return (self.fused and x.is_cuda and ops.curve_attn_supported(C, cn + cl)
                and ops.curve_agg_lds_bytes(cn, cl, C, C // 2) <= ops.CURVE_AGG_LDS_LIMIT)

    def forward(self, x, curves, cl=False):
        """Reference layout (default): x [B,C,N], curves [B,C,cn,cl] -> [B,C,N]; cl=True: x [B,N,C], curves
        [B,cn,cl,C] -> [B,N,C]."""
        if cl:
            B, cn, clen, C = curves.shape
            if self._kernel_form(x, cn, clen, C):
                return self._forward_kv(x, curves)
            return _cl(self.forward_steps(_cf(x), curves.permute(0, 3, 1, 2)))
        B, C, cn, clen = curves.shape
        if self._kernel_form(x, cn, clen, C):
            return _cf(self._forward_kv(_cl(x), curves.permute(0, 2, 3, 1)))
        return self.forward_steps(x, curves)

    def _forward_kv(self, x, curves):
        """Two launches: the curves become attention keys / values with convc and convd already applied
        (pc3d_curve_agg_kv_f32); per point that leaves  leaky(x + softmax(x^T K_inter) V_inter + softmax(x^T K_intra)
        V_intra) (pc3d_curve_attn_f32). x [B,N,C], curves [B,cn,cl,C]."""
        cn = curves.shape[1]
        fold = lambda m: folded_pw(m)[2:]                                            # noqa: E731
        (w_att, _), (wa, _), (wb, _), (wc, _) = fold(self.line_conv_att), fold(self.conva), fold(self.convb), fold(self.convc)
        (wn, _), (wl, _), (wd, bd) = fold(self.convn), fold(self.convl), fold(self.convd)
        Kp, Vp = ops.curve_agg_kv(curves, w_att, wa, wb, wn, wl, wc, wd, bd)
        return ops.curve_attn(x, Kp, Vp, cn, 0.2)

    def forward_steps(self, x, curves):
        """The same block as the reference's sequence of 1x1 convs, softmaxes and products (:393-437)."""
        att = pw(self.line_conv_att, curves)                                          # [B,1,cn,cl]
        inter = pw(self.conva, torch.sum(curves * F.softmax(att, dim=-1), dim=-1))    # [B,mid,cn]
        intra = pw(self.convb, torch.sum(curves * F.softmax(att, dim=-2), dim=-2))    # [B,mid,cl]
        q = pw(self.convc, x).transpose(1, 2).contiguous()                            # [B,N,mid]
        w_inter = F.softmax(torch.bmm(q, inter), dim=-1)                          # [B,N,cn]
        w_intra = F.softmax(torch.bmm(q, intra), dim=-1)                          # [B,N,cl]
        f_inter = torch.bmm(w_inter, pw(self.convn, inter).transpose(1, 2).contiguous())
        f_intra = torch.bmm(w_intra, pw(self.convl, intra).transpose(1, 2).contiguous())
        fused = torch.cat((f_inter, f_intra), dim=-1).transpose(1, 2).contiguous()
        return F.leaky_relu(x + pw(self.convd, fused), negative_slope=0.2)


class CurveGrouping(nn.Module):
    """:440-466 — pick curve_num start points by self-attention score, then walk."""

    def __init__(self, in_channel, k, curve_num, curve_length):
        super(CurveGrouping, self).__init__()
        self.curve_num = curve_num
        self.curve_length = curve_length
        self.in_channel = in_channel
        self.k = k
        self.att = nn.Conv1d(in_channel, 1, kernel_size=1, bias=False)
        self.walk = Walk(in_channel, k, curve_num, curve_length)

    def forward(self, x, xyz, idx, cl=False):
        """Reference layout (default): x [B,C,N], idx [B,N,k] -> curves [B,C,cn,cl]; cl=True: x [B,N,C], idx int32
        [B,N,k] -> [B,cn,cl,C].
        The reference asks topk for sorted=False (:457): the order of the start points is then unspecified (torch's CPU
        and GPU kernels return different ones) — yet the walk's momentum step mixes values of DIFFERENT curves by
        position (walk.py:104-105), so the output depends on it. The mirror fixes the order to descending score, ties
        to the lower index (a valid instance of "unsorted"), which makes the forward reproducible and lets fixtures pin
        it."""
        if not cl:
            return self.forward(_cl(x), None, idx.to(torch.int32).contiguous(), cl=True).permute(0, 3, 1, 2)
        B, N, C = x.shape
        _, _, w, _ = folded_pw(self.att)
        if x.is_cuda and C % 4 == 0:
            xs, att = ops.att_scale(x, w)                          # sigmoid(att(x)) and x * that, one launch
        else:
            att = torch.sigmoid(x @ w.reshape(-1))
            xs = x * att.unsqueeze(-1)
        if x.is_cuda and N <= ops.TOPK_MAX_N:
            start = ops.topk_desc(att, self.curve_num)
        else:
            start = torch.sort(att.detach(), dim=1, descending=True, stable=True)[1][:, :self.curve_num].to(torch.int32)
        return self.walk(None, xs, idx, start, cl=True)              # [B,cn,cl,C]


class MaskedMaxPool(nn.Module):
    """:469-484."""

    def __init__(self, npoint, radius, k):
        super(MaskedMaxPool, self).__init__()
        self.npoint = npoint
        self.radius = radius
        self.k = k

    def forward(self, xyz, features, cl=False, geo=None):
        """Reference layout (default): xyz [B,N,3], features [B,C,N] -> ([B,S,3], [B,C,S]); cl=True: features [B,N,C]
        -> [B,S,C]. FPS (start index 0, :81) + ball query + max over each ball in one gather-max launch instead of the
        [B,S,k,C] grouped tensor. geo = (fps_idx, ball-query idx) when CurveNet._geometry has computed them ahead."""
        if not cl:
            sub, y = self.forward(xyz, _cl(features), cl=True)
            return sub, _cf(y)
        xyz = xyz.float()
        B = xyz.shape[0]
        if geo is None:
            hold_rng_position(B, xyz.shape[1])
            fps_idx = ops.fps(xyz, self.npoint, None)
        else:
            fps_idx, idx = geo
        sub_xyz = ops.group_gather(xyz, None, fps_idx.view(B, self.npoint, 1)).view(B, self.npoint, 3)
        if geo is None:
            idx = ops.ball_query(self.radius, self.k, xyz, sub_xyz)
        return sub_xyz, ops.gather_max_rows(features, idx)
