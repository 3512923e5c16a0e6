"""PointNet++ single-scale-grouping classifier — MI355X mirror of model/pointnet2_SSG.py:230-254 (``PointNet_Ssg``).
Same sub-module names / state_dict keys; sampling and grouping run on the HIP kernels of pointnet2_utils."""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

from .pointnet import _FrozenFusedMixin, _fold_bn, _plain
from .pointnet2_utils import PointNetSetAbstraction, geometry_chain, geometry_join


class PointNet_Ssg(_FrozenFusedMixin, nn.Module):
    sampling_chain_front = True

    def sampling_input_sizes(self, N):
        """Points each farthest-point-sampling layer of ONE forward over N input points draws its start index from, in call
        order (pointnet2_utils.PredrawnFpsStarts)."""
        return [int(N), int(self.sa1.npoint)]
   # the forward starts with an FPS chain: attacks overlap their own searches with it

    def __init__(self, num_classes=40):
        super(PointNet_Ssg, self).__init__()
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=3, mlp=[64, 64, 128], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], group_all=True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, num_classes)
        self._folded_cache = None

    def _fold(self):
        return (_fold_bn(self.fc1.weight, self.fc1.bias, self.bn1), _fold_bn(self.fc2.weight, self.fc2.bias, self.bn2),
                _plain(self.fc3.weight, self.fc3.bias))

    def forward(self, xyz):
        self._require_fused(xyz)
        B, _, _ = xyz.shape
        head = self.folded()
        geo = geometry_chain(self, xyz, (self.sa1, self.sa2))     # FPS + ball queries of both layers, on a side stream
        l1_xyz, l1_points = self.sa1(xyz, None, geo=geo[0])
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, geo=geo[1])
        geometry_join(self, xyz)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        x = l3_points.reshape(B, 1024)
        x = ops.head_mlp(x, [(*head[0], "relu", 0.0), (*head[1], "relu", 0.0), (*head[2], None, 0.0)])   # dropout: identity in eval
        x = F.log_softmax(x, -1)
        return x, x, x
