"""EXPERIMENT: ATen operators (with input shapes) inside ONE eager forward + backward of a victim: usage
aten_ops_victim.py <curvenet|dgcnn|ssg> [B] [N]."""
import importlib, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from helpers import unit_cloud
M = importlib.import_module
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "curvenet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else (4096 if which == "curvenet" else 1024)
sd = M("3dpointcloudattack_amd.seeding").seeded_state_dict
if which == "curvenet":
    net = M("3dpointcloudattack_amd.model.curvenet").CurveNet(num_classes=40)
elif which == "dgcnn":
    net = M("3dpointcloudattack_amd.model.dgcnn").DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
else:
    net = M("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg(40)
net.load_state_dict(sd(net, 0)); net = net.to(dev).eval()
rng = np.random.default_rng(0)
x0 = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).transpose(1, 2).contiguous().to(dev)
def fb():
    x = x0.clone().requires_grad_()
    out = net(x)[0]
    out.square().sum().backward()
for _ in range(3):
    fb()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as p:
    fb(); torch.cuda.synchronize()
rows = [(e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]) for e in p.key_averages(group_by_input_shape=True)
        if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(reverse=True)
print("ATen device us per forward+backward:", round(sum(r[0] for r in rows), 1), "(parents and children both listed)")
for t, n, k, sh in rows[:45]:
    print(f"{k:30s} n={n:3d} us={t:8.1f} {sh}")
