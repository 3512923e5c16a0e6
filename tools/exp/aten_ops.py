"""EXPERIMENT: ATen operators (not pc3d kernels) in one eager forward + backward-to-input of a victim, by device time.
usage: python tools/exp/aten_ops.py <dgcnn|curvenet|ssg>"""
import importlib, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
M = importlib.import_module
dev = torch.device("cuda:0")
seeded_state_dict = M("3dpointcloudattack_amd.seeding").seeded_state_dict
which = sys.argv[1] if len(sys.argv) > 1 else "dgcnn"
if which == "dgcnn":
    net = M("3dpointcloudattack_amd.model.dgcnn").DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
    B, N = 32, 1024
elif which == "curvenet":
    net = M("3dpointcloudattack_amd.model.curvenet").CurveNet(num_classes=40)
    B, N = 32, 4096
else:
    net = M("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg(num_classes=40)
    B, N = 64, 2048
net.load_state_dict(seeded_state_dict(net, 0)); net = net.to(dev).eval()
x = (torch.rand(B, 3, N, device=dev) - 0.5).requires_grad_()
for _ in range(2):
    net(x)[0].sum().backward()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    net(x)[0].sum().backward()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ev if e.key.startswith("aten::") and e.device_time_total > 0 and not e.key.startswith("aten::clone") and not e.key.startswith("aten::contiguous")]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print(which, "aten device us (leaf-ish ops):", round(tot, 1))
for e in rows[:22]:
    print(f"{e.key:30s} n={e.count:3d} dev_us={e.device_time_total:8.1f} shapes={str(e.input_shapes)[:120]}")
