"""Experiment: GeoA3's whole _forward_step (victim + all loss terms) as one pair of hipGraphs vs eager / victim-only graphs."""
import importlib, sys, os, json, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import unit_cloud
from test_oracle_golden import _geo_cfg
M = importlib.import_module
seeded_state_dict = M("3dpointcloudattack_amd.seeding").seeded_state_dict
graphed = M("3dpointcloudattack_amd.graphed")
ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
lu = M("3dpointcloudattack_amd.attack.GeoA3.loss_utils"); ut = M("3dpointcloudattack_amd.attack.GeoA3.utility")
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "dgcnn"
if which == "dgcnn":
    B, N = 32, 1024
    net = M("3dpointcloudattack_amd.model.dgcnn").DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
else:
    B, N = 32, 4096
    net = M("3dpointcloudattack_amd.model.curvenet").CurveNet(num_classes=40)
net.load_state_dict(seeded_state_dict(net, 0)); net = net.to(dev).eval()
rng = np.random.default_rng(0)
pc = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).transpose(1, 2).contiguous().to(dev)
cfg = _geo_cfg(npoint=N, cls_loss_type='CE', hd_loss_weight=0.1, curv_loss_weight=1.0)
normal = ut.estimate_normal(pc, k=3)
kappa = lu._get_kappa_ori(pc, normal, cfg.curv_loss_knn)
target = torch.randint(0, 40, (B,), device=dev)
scale = torch.full((B,), 10.0, device=dev)
def step(net_, x, sc, tg):
    out = ga._forward_step(net_, pc, x, normal, kappa, tg, sc, cfg, False)
    return out[2], out[0], out[3], out[8]
def run(fn):
    x = (pc + 0.001 * torch.randn_like(pc)).requires_grad_()
    loss = fn(x, scale, target)[0]
    loss.backward()
    return x.grad
def timeit(fn, n=10, warm=4):
    for _ in range(warm): run(fn)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run(fn)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {"eager_ms": timeit(lambda x, s, t: step(net, x, s, t))}
gv = graphed.wrap(net)
res["victim_graph_ms"] = timeit(lambda x, s, t: step(gv, x, s, t))
x0 = (pc + 0.001 * torch.randn_like(pc)).requires_grad_()
fg = torch.cuda.make_graphed_callables(lambda x, s, t: step(net, x, s, t), (x0, scale.clone(), target.clone()))
res["step_graph_ms"] = timeit(fg)
torch.manual_seed(0); g1 = run(lambda x, s, t: step(net, x, s, t)).clone()
torch.manual_seed(0); g2 = run(fg).clone()
res["grad_rel_diff"] = float((g1 - g2).norm() / g1.norm())
print(json.dumps(res))
