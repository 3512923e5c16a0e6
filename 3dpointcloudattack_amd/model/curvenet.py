"""CurveNet classifier — MI355X mirror of model/curvenet.py:11-73 (same module tree / state_dict keys; returns RAW
logits three times like the reference)."""
import torch
import torch.nn as nn

from .. import graphed as _graphed
from .. import ops
from .. import streams as _streams
from .curvenet_util import CIC, LPFA, folded_pw, hold_rng_position, pw_cl
from .pointnet import _FrozenFusedMixin

curve_config = {
    'default': [[100, 5], [100, 5], None, None],
    'long': [[10, 30], None, None, None]
}


class CurveNet(_FrozenFusedMixin, nn.Module):
    # forward is a pure function of its input: attack loops may share / replay it. Its only RNG use is the discarded
    # FPS start draw (curvenet_util.hold_rng_position), which a hipGraph replay re-issues through consume_forward_rng.
    deterministic_forward = True
    geometry_stream = True      # FPS / ball queries / kNN graphs on a side stream beside the feature path
    sampling_chain_front = True   # the forward starts with an FPS chain: attacks overlap their own searches with it

    def _blocks(self):
        return (self.cic11, self.cic12, self.cic21, self.cic22, self.cic31, self.cic32, self.cic41, self.cic42)

    def consume_forward_rng(self, x):
        """Advance torch's CPU generator exactly as one forward on x [B,3,N] does (one discarded draw per down-sampling
        block) — called by GraphedVictim instead of the Python forward it replays."""
        B, n = x.shape[0], x.shape[-1]
        for blk in self._blocks():
            if n != blk.npoint:
                hold_rng_position(B, n)
                n = blk.npoint

    def __init__(self, num_classes=40, k=20, setting='default'):
        super(CurveNet, self).__init__()
        assert setting in curve_config
        additional_channel = 32
        self.lpfa = LPFA(9, additional_channel, k=k, mlp_num=1, initial=True)
        cfg = curve_config[setting]
        # encoder
        self.cic11 = CIC(npoint=1024, radius=0.05, k=k, in_channels=additional_channel, output_channels=64, bottleneck_ratio=2, mlp_num=1, curve_config=cfg[0])
        self.cic12 = CIC(npoint=1024, radius=0.05, k=k, in_channels=64, output_channels=64, bottleneck_ratio=4, mlp_num=1, curve_config=cfg[0])
        self.cic21 = CIC(npoint=1024, radius=0.05, k=k, in_channels=64, output_channels=128, bottleneck_ratio=2, mlp_num=1, curve_config=cfg[1])
        self.cic22 = CIC(npoint=1024, radius=0.1, k=k, in_channels=128, output_channels=128, bottleneck_ratio=4, mlp_num=1, curve_config=cfg[1])
        self.cic31 = CIC(npoint=256, radius=0.1, k=k, in_channels=128, output_channels=256, bottleneck_ratio=2, mlp_num=1, curve_config=cfg[2])
        self.cic32 = CIC(npoint=256, radius=0.2, k=k, in_channels=256, output_channels=256, bottleneck_ratio=4, mlp_num=1, curve_config=cfg[2])
        self.cic41 = CIC(npoint=64, radius=0.2, k=k, in_channels=256, output_channels=512, bottleneck_ratio=2, mlp_num=1, curve_config=cfg[3])
        self.cic42 = CIC(npoint=64, radius=0.4, k=k, in_channels=512, output_channels=512, bottleneck_ratio=4, mlp_num=1, curve_config=cfg[3])
        self.conv0 = nn.Sequential(nn.Conv1d(512, 1024, kernel_size=1, bias=False), nn.BatchNorm1d(1024), nn.ReLU(inplace=True))
        self.conv1 = nn.Linear(1024 * 2, 512, bias=False)
        self.conv2 = nn.Linear(512, num_classes)
        self.bn1 = nn.BatchNorm1d(512)
        self.dp1 = nn.Dropout(p=0.5)
        self._folded_cache = None

    def _geometry(self, pos, with_grad=False):
        """Everything in a forward that depends on the COORDINATES only — the FPS chain, the ball queries of the
        down-sampling blocks and the kNN graph of every resolution (curvenet_util.py:69-113, :10-17) — for detached
        pos [B,N,3], on the current stream: one (pool, graph, pool event, graph event) entry per block. FPS is a chain of npoint dependent
        arg-max steps on one workgroup per cloud (0.9 ms for 4096 -> 1024 at B=32, on 32 of 256 CUs): forward() runs
        this on a side stream beside the feature path, which waits for each level's event where it needs it."""
        B = pos.shape[0]
        levels, graphs, pts = [], {}, pos
        for blk in self._blocks():
            pool = ev_pool = None
            if pts.shape[1] != blk.npoint:
                hold_rng_position(B, pts.shape[1])
                fps_idx = ops.fps(pts, blk.npoint, None)
                sub = ops.group_gather(pts, None, fps_idx.view(B, blk.npoint, 1)).view(B, blk.npoint, 3)
                pool = (fps_idx, ops.ball_query(blk.radius, blk.k, pts, sub))
                pts = sub
                ev_pool = torch.cuda.Event()        # the block's max-pool + first 1x1 layer can start here ...
                ev_pool.record()
            key = (pts.shape[1], blk.k)
            if key not in graphs:
                g = ops.knn_graph(pts, blk.k)       # (idx, idx[:, :, 1:], idx[:, :, :k]) from one launch
                ev = torch.cuda.Event()             # ... the walk and LPFA wait for the graph only
                ev.record()
                graphs[key] = (g, ev, pts.shape[1])
            levels.append((pool, graphs[key][0], ev_pool, graphs[key][1]))
        if with_grad:
            # the LPFA blocks of a resolution gather through the first k columns of its graph; their deterministic backward
            # gathers back through the sorted reverse index — read by the BACKWARD only, so it is built after everything the
            # forward waits for (round 4: inside the level loop it sat on the forward's critical path: 54 us at the first level)
            for g, _, n in graphs.values():
                ops.attach_rev_index(g[2], n)
        return levels

    def forward(self, xyz):
        self._require_fused(xyz)
        # channels-last from here on: every block is a chain of this library's launches on [B,N,C] rows
        pos = xyz.float().transpose(1, 2).contiguous()
        blocks = self._blocks()
        cur = torch.cuda.current_stream()
        with_grad = torch.is_grad_enabled() and xyz.requires_grad
        with torch.no_grad():
            if self.geometry_stream:
                side = _streams.side_stream(pos.device, _streams.GEOMETRY)      # ONE per process (see streams.py)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    levels = self._geometry(pos.detach(), with_grad)
            else:
                levels = self._geometry(pos.detach(), with_grad)
        feats = self.lpfa(None, pos, None, cl=True)
        _graphed.note_input_knn(self, xyz, self.lpfa.__dict__.get("_last_idx"))   # the input cloud's graph: attacks may reuse it
        for blk, geo in zip(blocks, levels):
            pos, feats = blk(pos, feats, cl=True, geo=geo)
        if self.geometry_stream:
            cur.wait_stream(side)          # join (every event above has been waited for; keeps captures well-formed)
        # conv0's ReLU is applied inside the pooling launch: [max_i relu(y) | mean_i relu(y)] (:66-68)
        _, _, w0, b0 = folded_pw(self.conv0, None)
        x = ops.linear_act_maxmean_pool(feats, w0, b0, 0.0)
        _, _, w1, b1 = folded_pw(self.conv1, self.bn1)
        _, _, w2, b2 = folded_pw(self.conv2, None)
        x = ops.head_mlp(x, [(w1, b1, "relu", 0.0), (w2, b2, None, 0.0)])      # (dropout is the identity in eval mode)
        return x, x, x
