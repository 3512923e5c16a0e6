"""PointNet++ multi-scale-grouping classifier — MI355X mirror of model/pointnet2_MSG.py:194-226 (``PointNet_Msg``),
used as a transfer model by the KNN attack (attack/KNN/KNN_attack.py:191)."""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

from .pointnet import _FrozenFusedMixin, _fold_bn, _plain
from .pointnet2_utils import PointNetSetAbstraction, PointNetSetAbstractionMsg, geometry_chain, geometry_join


class PointNet_Msg(_FrozenFusedMixin, nn.Module):
    sampling_chain_front = True

    def sampling_input_sizes(self, N):
        """Points each farthest-point-sampling layer of ONE forward over N input points draws its start index from, in call
        order (pointnet2_utils.PredrawnFpsStarts)."""
        return [int(N), int(self.sa1.npoint)]
   # the forward starts with an FPS chain: attacks overlap their own searches with it

    def __init__(self, num_class, normal_channel=True):
        super(PointNet_Msg, self).__init__()
        in_channel = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], in_channel, [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = PointNetSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320, [[64, 64, 128], [128, 128, 256], [128, 128, 256]])
        self.sa3 = PointNetSetAbstraction(None, None, None, 640 + 3, [256, 512, 1024], True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.5)
        self.fc3 = nn.Linear(256, num_class)
        self._folded_cache = None

    def _fold(self):
        return (_fold_bn(self.fc1.weight, self.fc1.bias, self.bn1), _fold_bn(self.fc2.weight, self.fc2.bias, self.bn2),
                _plain(self.fc3.weight, self.fc3.bias))

    def forward(self, xyz):
        self._require_fused(xyz)
        B, _, _ = xyz.shape
        if self.normal_channel:
            norm = xyz[:, 3:, :]
            xyz = xyz[:, :3, :]
        else:
            norm = None
        head = self.folded()
        geo = geometry_chain(self, xyz, (self.sa1, self.sa2))     # FPS + ball queries of both layers, on a side stream
        l1_xyz, l1_points = self.sa1(xyz, norm, geo=geo[0])
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, geo=geo[1])
        geometry_join(self, xyz)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        x = l3_points.reshape(B, 1024)
        x = ops.head_mlp(x, [(*head[0], "relu", 0.0), (*head[1], "relu", 0.0), (*head[2], None, 0.0)])
        x = F.log_softmax(x, -1)
        return x, x, x
