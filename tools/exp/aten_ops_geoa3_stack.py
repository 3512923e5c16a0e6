"""EXPERIMENT: ATen operators that launch a kernel inside the GeoA3 loop on DGCNN (B=32, N=1024), with the package frames
of their Python stacks: where each remaining non-own launch comes from."""
import importlib, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from helpers import unit_cloud
from test_oracle_golden import _geo_cfg
M = importlib.import_module
dev = torch.device("cuda:0")
seeded_state_dict = M("3dpointcloudattack_amd.seeding").seeded_state_dict
net = M("3dpointcloudattack_amd.model.dgcnn").DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
net.load_state_dict(seeded_state_dict(net, 0)); net = net.to(dev).eval()
rng = np.random.default_rng(0)
B, N = 32, 1024
pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
with torch.no_grad():
    lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
def run(it):
    cfg = _geo_cfg(iter_max_steps=it, binary_max_steps=1, npoint=N, cls_loss_type='CE', hd_loss_weight=0.1, curv_loss_weight=1.0)
    torch.manual_seed(0); np.random.seed(0)
    ga.geoA3_attack(net, None, None, None, None, None, pcs, lab, cfg, 0, 1)
run(4)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as p:
    run(6); torch.cuda.synchronize()
rows = {}
for e in p.key_averages(group_by_input_shape=True, group_by_stack_n=14):
    if not e.key.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    frames = [f for f in e.stack if "3dpointcloudattack_amd" in f or "attack" in f]
    key = (e.key, str(e.input_shapes)[:60], " <- ".join(f.split("3dpointcloudattack_amd/")[-1][:70] for f in frames[:3]))
    n, t = rows.get(key, (0, 0.0))
    rows[key] = (n + e.count, t + e.self_device_time_total)
out = sorted(((t, n, k) for k, (n, t) in rows.items()), reverse=True)
print("ops with own device time (6 iterations + setup):")
for t, n, k in out[:60]:
    print(f"{k[0]:26s} n={n:4d} us={t:8.1f} {k[1]:60s} {k[2]}")
