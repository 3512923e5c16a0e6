"""EXPERIMENT: ATen operators per iteration of the CW loop on CurveNet (B=32, N=4096; graphed victim), by device time."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from helpers import unit_cloud
M = importlib.import_module
dev = torch.device("cuda:0")
seeded_state_dict = M("3dpointcloudattack_amd.seeding").seeded_state_dict
net = M("3dpointcloudattack_amd.model.curvenet").CurveNet(num_classes=40)
net.load_state_dict(seeded_state_dict(net, 0)); net = net.to(dev).eval()
rng = np.random.default_rng(0)
B, N = 32, 4096
pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
with torch.no_grad():
    lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
cw = M("3dpointcloudattack_amd.attack.CW.CW_attack")
adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
def run(it):
    atk = cw.CW(net, net, adv.UntargetedLogitsAdvLoss(kappa=0.), cu.ClipPointsLinf(budget=0.18), du.ChamferDist(method='adv2ori'),
                attack_lr=1e-2, binary_step=1, num_iter=it, graph=True)
    atk.attack(pcs, lab)
run(4)
def prof(it):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as p:
        run(it); torch.cuda.synchronize()
    return {(e.key, str(e.input_shapes)[:100]): (e.count, e.device_time_total) for e in p.key_averages(group_by_input_shape=True)
            if e.key.startswith("aten::") and e.device_time_total > 0}
a, b = prof(2), prof(6)
rows = [((t1 - a.get(k, (0, 0.0))[1]) / 4, (n1 - a.get(k, (0, 0.0))[0]) / 4, k) for k, (n1, t1) in b.items() if n1 > a.get(k, (0, 0.0))[0]]
rows.sort(reverse=True)
for t, n, k in rows[:30]:
    print(f"{k[0]:28s} n/it={n:5.1f} us/it={t:7.1f} {k[1]}")
