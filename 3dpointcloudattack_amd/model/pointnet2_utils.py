"""PointNet++ sampling / grouping ops and set-abstraction modules — MI355X mirror of model/pointnet2_utils.py
(same function names, argument order, return layouts, ``state_dict`` keys ``mlp_convs.N`` / ``mlp_bns.N`` /
``conv_blocks.i.j`` / ``bn_blocks.i.j``).

What runs where: farthest_point_sample, query_ball_point, index_points and the grouped gather (with its backward — the
attack's gradient path through the grouping) are single HIP launches (pc3d_fps_f32, pc3d_ball_query_f32,
pc3d_group_gather_f32). The reference's versions are a Python loop of npoint x 6 launches, a full sort of a [B,S,N]
int64 tensor and advanced-indexing gathers. The grouped 1x1-conv MLPs run channels-last (no permute / contiguous
copies) on the hand-written fp32-MFMA kernels (pc3d_gemm_nt_f32 for layers 1-2, pc3d_group_linear_max_f32 for the last
layer + max); eval-mode BatchNorm2d is folded into the conv weights.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import streams as _streams
from .pointnet import _FrozenFusedMixin



def _mlp_max(x, layers):
    """[B,S,ns,C0] -> [B,S,C_last]: the shared MLP of a set-abstraction layer and the max over the group; the last
    layer + max use the sparse-backward operator when the group fits it."""
    if x.shape[2] <= ops.GROUP_MAX_NS and x.is_cuda:
        return ops.mlp_relu_max(x, layers)
    for w, b in layers:
        x = _linear_relu(x, w, b)
    return torch.max(x, 2)[0]


def _split_first(layers):
    """(Wx [C1,3], Wf [C1,D] or None, b1) of a folded first layer in the kernels' [xyz(3), feat(D)] input order."""
    w1, b1 = layers[0]
    return w1[:, :3].contiguous(), (w1[:, 3:].contiguous() if w1.shape[1] > 3 else None), b1


FUSE_LAYERS_1_2 = True    # layer 1 generated inside layer 2's GEMM (ops.grouped_mlp_max); False: group_act + mlp_relu_max
SA_FRONT = True           # a single-scale layer's centres / per-point / per-centre first layer as one node (ops.sa_front)
SPLIT_GROUP_ALL = True    # the group-all layer without the [xyz ; points] concatenation
REVERSE_INDEX = True      # the geometry chain also builds the reverse index of every grouping (ops.group_reverse): layer 1's
                          # backward gathers through it; False: scatter with float atomics


def _front_supported(first, layers, ns):
    """Shapes the per-point / per-centre form of the first layer takes (ops.group_act / ops.grouped_mlp_max behind it)."""
    C1 = first[0].shape[0]
    return C1 % 4 == 0 and C1 <= ops.GROUP_ACT_MAX_C and len(layers) >= 2 and ns <= ops.GROUP_MAX_NS


def _grouped_tail(P, Bc, idx, layers, rev=None, blocks=None):
    """relu(P[idx] + Bc) -> the remaining layers -> max over the group: [B,S,C_last]."""
    C1 = P.shape[2]
    if FUSE_LAYERS_1_2 and ops.grouped_mlp_max_supported(C1, idx.shape[2], layers[1:]):
        return ops.grouped_mlp_max(P, Bc, idx, layers[1:], rev=rev, blocks=blocks)   # layer 1 generated inside layer 2's GEMM
    h1 = ops.group_act(P, Bc, idx, 0.0)                               # [B,S,ns,C1] = relu(layer 1)
    return ops.mlp_relu_max(h1, layers[1:])


def _grouped_mlp_max(xyz_t, pts, idx, fps_idx, layers, first, rev=None):
    """The shared MLP + group max of one grouping scale, first layer WITHOUT the grouped input tensor:
    W1 [x_j - c_s ; f_j] + b1 = P[idx[s,j]] + Bc[s], P = Wx x + Wf f per POINT (B*N rows instead of B*S*ns: 16x fewer
    at SSG's second layer), Bc = b1 - (Wx x)[centroid]; pc3d_group_act_f32 gathers P and applies the ReLU, i.e. it
    emits the layer-1 OUTPUT where the reference (and round 1) gathered the layer-1 INPUT and ran a [B*S*ns, 3+D]
    GEMM on it (model/pointnet2_utils.py:118-135,190-197). (The multi-scale layers' form: the single-scale layer takes
    P, Bc and the centres from ops.sa_front in one autograd node.)"""
    wx, wf, b1 = first
    C1 = wx.shape[0]
    B, N, _ = xyz_t.shape
    S = fps_idx.shape[1]
    if not _front_supported(first, layers, idx.shape[2]):
        # widths the gather-with-activation kernel does not take: the plain form (gather the grouped input, then the MLP)
        new_xyz = ops.group_gather(xyz_t, None, fps_idx.view(B, S, 1)).view(B, S, 3)
        g = ops.group_gather(xyz_t, pts, idx, centers=new_xyz.detach(), center_idx=fps_idx)
        return _mlp_max(g, layers)
    px = ops.affine3(xyz_t, wx)                                       # [B,N,C1] = Wx x, xyz read through its strides
    P = px if pts is None else ops.linear_res_act(pts, wf, None, px)
    Bc = b1 - ops.group_gather(None, px, fps_idx.view(B, S, 1)).view(B, S, C1)
    return _grouped_tail(P, Bc, idx, layers, rev)


def _linear_relu(x, w, b):
    """relu(x @ w.T + b) over the last dimension on the fp32-MFMA point-wise kernel (pc3d_gemm_nt_f32: bias + ReLU in
    the epilogue; its backward applies the ReLU mask while loading dY, so neither direction makes a separate
    activation pass over the [B*S*ns, C] tensor)."""
    return ops.linear_act(x, w, b, "relu")


def pc_normalize(pc):
    """model/pointnet2_utils.py:11-17."""
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    return pc / m


def square_distance(src, dst):
    """:19-38 — [B,N,C=3],[B,M,3] -> [B,N,M] squared distances (dense; direct-difference form)."""
    return ops.pairwise(src.float(), dst.float())


def _i32(idx):
    return idx if idx.dtype == torch.int32 else idx.to(torch.int32)


def index_points(points, idx):
    """:41-57 — points [B,N,C], idx [B,S] or [B,S,ns] -> [B,S,C] / [B,S,ns,C]. Differentiable in points."""
    points = points.float()
    if idx.dim() == 2:
        B, S = idx.shape
        return ops.group_gather(None, points, _i32(idx).reshape(B, S, 1).contiguous()).view(B, S, -1)
    return ops.group_gather(None, points, _i32(idx).contiguous())


_FPS_START_SOURCE = None      # None: the global CPU generator, like the reference; else callable(B, N, device) -> int32 [B]


def set_fps_start_source(src):
    """Where farthest-point sampling takes its start indices from: None = one torch.randint(0, N, (B,)) on the global CPU
    generator per sampling layer and forward (model/pointnet2_utils.py:72, SURVEY A-4), or a callable (B, N, device) ->
    int32 [B] device tensor (SeededFpsStarts: a stream per SAMPLE, what a sharded run needs). Returns the previous one."""
    global _FPS_START_SOURCE
    prev, _FPS_START_SOURCE = _FPS_START_SOURCE, src
    return prev


class SeededFpsStarts:
    """FPS start indices drawn per SAMPLE: sample i takes its r-th start from its own generator (seed seeds[i]), whatever
    batch or shard it is evaluated in — the reference's single shared stream (:72) makes a sample's starts depend on its
    position in the batch and on every other forward of the process ("FPS start indices must be drawn
    per-sample from a seed, not from a shared stream")."""

    def __init__(self, seeds):
        self.gens = [torch.Generator().manual_seed(int(s)) for s in seeds]

    def draw_cpu(self, B, N):
        if B != len(self.gens):
            raise ValueError(f"SeededFpsStarts: batch of {B} clouds, {len(self.gens)} seeds")
        return torch.tensor([int(torch.randint(0, N, (1,), generator=g)) for g in self.gens], dtype=torch.int32)

    def __call__(self, B, N, device):
        return ops.h2d(self.draw_cpu(B, N), device, torch.int32)


def _draw_cpu(source, B, N):
    """One sampling layer's start indices on the host: the source's own draw, or the reference's — the first index from the
    GLOBAL CPU generator (:72, SURVEY A-4); same call, same stream position -> same start indices under the same seed."""
    if source is not None:
        return source.draw_cpu(B, N)
    return torch.randint(0, N, (B,), dtype=torch.long).to(torch.int32)


class PredrawnFpsStarts:
    """The start indices of `forwards` consecutive forwards of ONE victim, drawn ahead — from the source the live draws
    would have used, in the order they would have been made: per forward one draw of B indices per sampling layer, layer l
    over `layer_sizes[l]` points — and uploaded ONCE. An attack loop's forwards then read theirs from the device buffer: no
    host draw and no host->device copy per sampling layer and forward (two per iteration for the single-scale classifier,
    each a stream operation the launch queue has to thread through). When the buffer is used up the previous source is
    live again at exactly the stream position it would have had; a call that does not fit the recorded pattern before that
    is an error (the draws made ahead cannot be returned to the generator)."""

    def __init__(self, layer_sizes, B, forwards, device, inner=None):
        self.sizes, self.B, self.inner = [int(n) for n in layer_sizes], int(B), inner
        rows = [_draw_cpu(inner, self.B, n) for _ in range(int(forwards)) for n in self.sizes]
        self.total = len(rows)
        self.buf = ops.h2d(torch.stack(rows), device, torch.int32) if rows else None      # [forwards * layers, B]
        self.pos = 0

    def draw_cpu(self, B, N):
        if self.pos < self.total:
            raise RuntimeError("PredrawnFpsStarts: host draw requested while pre-drawn indices are pending")
        return _draw_cpu(self.inner, B, N)

    def __call__(self, B, N, device):
        if self.pos >= self.total:
            return ops.h2d(_draw_cpu(self.inner, B, N), device, torch.int32)
        if B != self.B or N != self.sizes[self.pos % len(self.sizes)] or self.buf.device != torch.device(device):
            raise RuntimeError(f"PredrawnFpsStarts: call (B={B}, N={N}) does not fit the pre-drawn pattern "
                               f"(B={self.B}, N={self.sizes[self.pos % len(self.sizes)]})")
        row = self.buf[self.pos]
        self.pos += 1
        return row


def fps_start_source():
    return _FPS_START_SOURCE


def _fps_start(B, N, device):
    if _FPS_START_SOURCE is not None:
        return _FPS_START_SOURCE(B, N, device)
    return ops.h2d(_draw_cpu(None, B, N), device, torch.int32)     # no queue stall (ops.h2d)


def farthest_point_sample(xyz, npoint, start=None):
    """:60-81 — xyz [B,N,3] -> centroids [B,npoint] int64. start (optional int tensor [B]) overrides the random draw."""
    B, N, _ = xyz.shape
    st = _fps_start(B, N, xyz.device) if start is None else start.to(device=xyz.device, dtype=torch.int32)
    return ops.fps(xyz.float(), npoint, st).long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    """:84-104 — [B,S,nsample] int64."""
    return ops.ball_query(radius, nsample, xyz.float(), new_xyz.float()).long()


def _sample_and_group_i32(npoint, radius, nsample, xyz, points):
    B, N, C = xyz.shape
    fps_idx = ops.fps(xyz, npoint, _fps_start(B, N, xyz.device))                      # [B,S] i32
    new_xyz = ops.group_gather(xyz, None, fps_idx.view(B, npoint, 1)).view(B, npoint, 3)
    idx = ops.ball_query(radius, nsample, xyz, new_xyz)                              # [B,S,ns] i32
    # [B,S,ns,3+D] = [xyz[idx]-new_xyz, points[idx]]; the centre's gradient is folded into xyz's inside the kernel
    new_points = ops.group_gather(xyz, points, idx, centers=new_xyz.detach(), center_idx=fps_idx)
    return new_xyz, new_points, idx, fps_idx


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """:107-135 — new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D]."""
    xyz = xyz.float()
    points = None if points is None else points.float()
    new_xyz, new_points, idx, fps_idx = _sample_and_group_i32(npoint, radius, nsample, xyz, points)
    if returnfps:
        grouped_xyz = ops.group_gather(xyz, None, idx)
        return new_xyz, new_points, grouped_xyz, fps_idx.long()
    return new_xyz, new_points


_ZEROS = {}


def _zeros_cached(B, S, C, device):
    """The all-zero centre of a group-all layer (:143): one tensor per (shape, device) instead of a fill per forward.
    Read-only by convention (the classifiers never look at it)."""
    key = (B, S, C, device.type, device.index)
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 16:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros(B, S, C, device=device)
    return z


def sample_and_group_all(xyz, points):
    """:138-155 — new_xyz zeros [B,1,3], new_points [B,1,N,3+D]."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped_xyz = xyz.view(B, 1, N, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points


def _fold_bn2d(conv, bn, perm=None):
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1).float()
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(s)
    w = w * s[:, None]
    if perm is not None:
        w = w[:, perm]
    return w.contiguous(), ((b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()).contiguous()


def _cl(points):
    """[B,D,N] channels-first API tensor -> [B,N,D] contiguous (free when it is a view of channels-last storage)."""
    return None if points is None else points.permute(0, 2, 1).contiguous().float()


class PointNetSetAbstraction(_FrozenFusedMixin, nn.Module):
    """:158-199."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super(PointNetSetAbstraction, self).__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.group_all = group_all
        self._folded_cache = None

    def _fold(self):
        layers = [_fold_bn2d(c, b) for c, b in zip(self.mlp_convs, self.mlp_bns)]
        return layers, _split_first(layers)

    def geometry(self, xyz_t, with_rev=True):
        """What this layer derives from the COORDINATES alone, for detached xyz_t [B,N,3] on the current stream: [centre
        indices [B,S] i32, detached centres [B,S,3], group indices [B,S,ns] i32, event, reverse index, its event]. A
        classifier runs the whole chain of its layers on a side stream (geometry_chain) beside the MLPs of the previous
        layer; with_rev=False leaves the last two entries None for geometry_rev to fill in later."""
        B, N, _ = xyz_t.shape
        fps_idx = ops.fps(xyz_t, self.npoint, _fps_start(B, N, xyz_t.device))                       # [B,S] i32
        centres = ops.group_gather(xyz_t, None, fps_idx.view(B, self.npoint, 1)).view(B, self.npoint, 3)
        idx = ops.ball_query(self.radius, self.nsample, xyz_t, centres)                              # [B,S,ns] i32
        ev = torch.cuda.Event()
        ev.record()                      # what the layer's FORWARD waits for (its front: centres and group indices)
        # which 8- / 16- / 32-row units of the grouped rows hold listed points at all (the chain launch packs those; ops.sa_blocks):
        # needed by the chain launch only, which waits for the table's own event after the layer's front has been queued
        blocks = None
        if self.nsample in (32, 64, 128) and len(self.mlp_convs) == 3:
            unit = ops.sa_chain_table_unit(self.npoint, self.nsample, *[c.out_channels for c in self.mlp_convs])
            tbl = ops.sa_blocks(idx, unit) if unit else None          # (None: too many groups for the packing launches)
            if tbl is not None:
                ev_tb = torch.cuda.Event()
                blocks = tbl + (ev_tb,)
                ev_tb.record()
        g = [fps_idx, centres, idx, ev, None, None, N, blocks]
        if with_rev:
            self.geometry_rev(g)
        return g

    @staticmethod
    def geometry_rev(g):
        """The reverse index of the grouping (for the backward of layer 1: no float atomics) + its event, on the current
        stream. Five launches (~85 us at SSG's first level) the layer's forward does not wait for: the consumer's stream
        waits for them AFTER it has queued the layer's MLP, and geometry_chain queues them after the sampling of every
        layer so that the next layer's centres are not held up either."""
        if REVERSE_INDEX:
            off, lst = ops.group_reverse(g[2], g[6])
            g[5] = torch.cuda.Event()
            g[5].record()
            g[4] = (off, lst, g[5])      # the backward that reads the index waits for the event (ops._GroupedMLPMaxFn)

    def forward(self, xyz, points, geo=None):
        """xyz [B,3,N], points [B,D,N] or None -> new_xyz [B,3,S], new_points [B,D',S]. geo: this layer's entry of
        geometry_chain (indices computed ahead on a side stream), or None to sample and group here."""
        self._require_fused(xyz)
        xyz_t = xyz.permute(0, 2, 1).float()      # strided view; the kernels take strides
        pts = _cl(points)
        layers, first = self.folded()
        if self.group_all:
            wx, wf, b1 = first
            if pts is not None and xyz_t.is_cuda and wx.shape[0] % 4 == 0 and len(layers) >= 2 and SPLIT_GROUP_ALL:
                # [xyz ; points] is never concatenated (:138-155): layer 1 = relu(Wf f + (Wx x + b1)), the coordinate
                # columns as the GEMM's residual operand
                B, N, _ = xyz_t.shape
                new_xyz = _zeros_cached(B, 1, 3, xyz_t.device)
                h1 = ops.linear_res_act(pts, wf, None, ops.affine3(xyz_t, wx, b1), "relu")
                new_points = _mlp_max(h1.view(B, 1, N, -1), layers[1:])
            else:
                new_xyz, new_points = sample_and_group_all(xyz_t, pts)
                new_points = _mlp_max(new_points, layers)   # [B,1,D'] channels-last 1x1 convs, no permutes
        else:
            B, N, _ = xyz_t.shape
            rev = None
            ev_rev = blocks = None
            if geo is not None:
                fps_idx, _, idx, ev, rev, ev_rev = geo[:6]
                blocks = geo[7] if len(geo) > 7 else None
                torch.cuda.current_stream(xyz_t.device).wait_event(ev)
            else:
                fps_idx = ops.fps(xyz_t, self.npoint, _fps_start(B, N, xyz_t.device))                   # [B,S] i32
                idx = None
            if SA_FRONT and xyz_t.is_cuda and _front_supported(first, layers, self.nsample):
                # centres, per-point and per-centre forms of layer 1 in ONE autograd node (ops.sa_front)
                new_xyz, P, Bc = ops.sa_front(xyz_t, pts, fps_idx, *first)
                if idx is None:
                    idx = ops.ball_query(self.radius, self.nsample, xyz_t, new_xyz.detach())
                new_points = _grouped_tail(P, Bc, idx, layers, rev, blocks)
            else:
                new_xyz = ops.group_gather(xyz_t, None, fps_idx.view(B, self.npoint, 1)).view(B, self.npoint, 3)
                if idx is None:
                    idx = ops.ball_query(self.radius, self.nsample, xyz_t, new_xyz)                      # [B,S,ns] i32
                new_points = _grouped_mlp_max(xyz_t, pts, idx, fps_idx, layers, first, rev=rev)
        return new_xyz.permute(0, 2, 1), new_points.permute(0, 2, 1)


class PointNetSetAbstractionMsg(_FrozenFusedMixin, nn.Module):
    """:202-259 — multi-scale grouping; per scale the reference concatenates [features, xyz] (features FIRST)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super(PointNetSetAbstractionMsg, self).__init__()
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        self.in_channel = in_channel
        for i in range(len(mlp_list)):
            convs = nn.ModuleList()
            bns = nn.ModuleList()
            last_channel = in_channel + 3
            for out_channel in mlp_list[i]:
                convs.append(nn.Conv2d(last_channel, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last_channel = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)
        self._folded_cache = None

    def _fold(self):
        D = self.in_channel
        # kernel layout is [xyz(3), feat(D)]; the reference's first conv expects [feat(D), xyz(3)]
        perm = torch.cat([torch.arange(D, D + 3), torch.arange(0, D)]) if D > 0 else None
        out = []
        for convs, bns in zip(self.conv_blocks, self.bn_blocks):
            layers = []
            for j, (c, b) in enumerate(zip(convs, bns)):
                layers.append(_fold_bn2d(c, b, perm.to(c.weight.device) if (j == 0 and perm is not None) else None))
            out.append((layers, _split_first(layers)))
        return out

    def geometry(self, xyz_t, with_rev=True):
        """[centre indices, detached centres, [group indices per scale], event, [reverse indices], their event] for
        detached xyz_t [B,N,3]; see PointNetSetAbstraction.geometry."""
        B, N, _ = xyz_t.shape
        S = self.npoint
        fps_idx = ops.fps(xyz_t, S, _fps_start(B, N, xyz_t.device))
        centres = ops.group_gather(xyz_t, None, fps_idx.view(B, S, 1)).view(B, S, 3)
        idxs = [ops.ball_query(radius, self.nsample_list[i], xyz_t, centres) for i, radius in enumerate(self.radius_list)]
        ev = torch.cuda.Event()
        ev.record()
        g = [fps_idx, centres, idxs, ev, None, None, N]
        if with_rev:
            self.geometry_rev(g)
        return g

    @staticmethod
    def geometry_rev(g):
        if REVERSE_INDEX:
            revs = [ops.group_reverse(ix, g[6]) for ix in g[2]]
            g[5] = torch.cuda.Event()
            g[5].record()
            g[4] = [(off, lst, g[5]) for off, lst in revs]

    def forward(self, xyz, points, geo=None):
        self._require_fused(xyz)
        xyz_t = xyz.permute(0, 2, 1).float()
        pts = _cl(points)
        B, N, C = xyz_t.shape
        S = self.npoint
        revs = ev_rev = None
        if geo is not None:
            fps_idx, _, idxs, ev, revs, ev_rev = geo[:6]
            torch.cuda.current_stream(xyz_t.device).wait_event(ev)
        else:
            fps_idx = ops.fps(xyz_t, S, _fps_start(B, N, xyz_t.device))
            idxs = None
        new_xyz = ops.group_gather(xyz_t, None, fps_idx.view(B, S, 1)).view(B, S, 3)
        outs = []
        for i, radius in enumerate(self.radius_list):
            idx = idxs[i] if idxs is not None else ops.ball_query(radius, self.nsample_list[i], xyz_t, new_xyz)
            layers, first = self.folded()[i]
            outs.append(_grouped_mlp_max(xyz_t, pts, idx, fps_idx, layers, first, rev=revs[i] if revs is not None else None))
        return new_xyz.permute(0, 2, 1), torch.cat(outs, dim=-1).permute(0, 2, 1)


def geometry_chain(owner, xyz, layers):
    """Sampling and grouping of consecutive set-abstraction layers on a SIDE stream: layer l+1 samples the centres of
    layer l, so the whole chain depends on the input coordinates only (model/pointnet2_utils.py:158-176 draws and
    searches inside each layer's forward). Farthest-point sampling is a chain of dependent arg-max steps on one
    workgroup per cloud; run ahead, the sampling and ball queries of layer l+1 overlap the MLP of layer l. The FPS start
    indices are drawn from the CPU generator in the same order as in-line. Returns one geometry() entry per layer;
    `owner.geometry_stream = False` (or a CPU tensor) computes them in-line on the current stream."""
    if not xyz.is_cuda:
        return [None] * len(layers)
    pts = xyz.detach().permute(0, 2, 1).float()
    cur = torch.cuda.current_stream(xyz.device)
    side = None
    if getattr(owner, "geometry_stream", True):
        side = _streams.side_stream(xyz.device, _streams.GEOMETRY)      # ONE per process (see streams.py)
        side.wait_stream(cur)
    out = []
    with torch.no_grad(), torch.cuda.stream(side if side is not None else cur):
        for layer in layers:             # the sampling chain first: layer l + 1 needs only the centres of layer l ...
            g = layer.geometry(pts, with_rev=False)
            out.append(g)
            pts = g[1]
        for layer, g in zip(layers, out):   # ... then the reverse indices, which only the backward reads
            layer.geometry_rev(g)
    return out


def geometry_join(owner, xyz):
    """The forward's last word to the side stream (keeps graph captures well-formed: every fork is joined)."""
    if xyz.is_cuda and getattr(owner, "geometry_stream", True):
        torch.cuda.current_stream(xyz.device).wait_stream(_streams.side_stream(xyz.device, _streams.GEOMETRY))


class PointNetFeaturePropagation(_FrozenFusedMixin, nn.Module):
    """:262-312 — 3-NN inverse-distance interpolation + MLP (segmentation head; not used by the attacks)."""

    def __init__(self, in_channel, mlp):
        super(PointNetFeaturePropagation, self).__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel
        self._folded_cache = None

    def _fold(self):
        return [_fold_bn2d(c, b) for c, b in zip(self.mlp_convs, self.mlp_bns)]

    def forward(self, xyz1, xyz2, points1, points2):
        self._require_fused(xyz1)
        x1 = xyz1.permute(0, 2, 1).float()
        x2 = xyz2.permute(0, 2, 1).float()
        p2 = _cl(points2)
        B, N, _ = x1.shape
        S = x2.shape[1]
        if S == 1:
            interpolated = p2.repeat(1, N, 1)
        else:
            dists, idx = ops.knn(x1, x2, 3)                       # the reference sorts [B,N,S] and keeps 3
            recip = 1.0 / (dists + 1e-8)
            weight = recip / torch.sum(recip, dim=2, keepdim=True)
            interpolated = torch.sum(ops.group_gather(None, p2, idx) * weight.view(B, N, 3, 1), dim=2)
        new_points = interpolated if points1 is None else torch.cat([_cl(points1), interpolated], dim=-1)
        for w, b in self.folded():
            new_points = _linear_relu(new_points, w, b)
        return new_points.permute(0, 2, 1)
