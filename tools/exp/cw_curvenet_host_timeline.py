"""EXPERIMENT: where the HOST spends an iteration of the CW loop on CurveNet (B=32, N=4096; graphed victim). No syncs are
added: host durations of the pieces of CW._iterate, against the loop's wall time per iteration."""
import importlib, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
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
cw = M("3dpointcloudattack_amd.attack.CW.CW_attack"); gr = M("3dpointcloudattack_amd.graphed")
adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
T = collections.defaultdict(list)
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            T[name].append(time.perf_counter() - t0)
    return w
MARK = []
_it = cw.CW._iterate
def _marked(self, st, *a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); MARK.append((time.perf_counter(), e))
    return _it(self, st, *a, **k)
cw.CW._iterate = timed("iterate", _marked)
gr.GraphedVictim.forward = timed("victim.forward", gr.GraphedVictim.forward)
gr.GraphedVictim._weights_key = timed("weights_key", gr.GraphedVictim._weights_key)
torch.Tensor.backward = timed("backward", torch.Tensor.backward)
def run(it):
    atk = cw.CW(net, net, adv.UntargetedLogitsAdvLoss(kappa=0.), cu.ClipPointsLinf(budget=0.18), du.ChamferDist(method='adv2ori'),
                attack_lr=1e-2, binary_step=1, num_iter=it, graph=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    atk.attack(pcs, lab)
    torch.cuda.synchronize(); return time.perf_counter() - t0
run(6); a = run(6); T.clear(); MARK.clear(); b = run(46)
h0, e0 = MARK[0]
for i in (1, 2, 3, 5, 10, 20, 30, 40, 45):
    h, e = MARK[i]
    print(f"iteration {i:2d}: host enters at {(h - h0) * 1e3:8.2f} ms, GPU reaches it at {e0.elapsed_time(e):8.2f} ms")
print(f"wall per iteration {(b - a) / 40 * 1e3:.2f} ms")
for k, v in T.items():
    v = v[-30:]
    print(f"host {k:16s} n={len(v):3d} mean {sum(v) / len(v) * 1e3:7.3f} ms  max {max(v) * 1e3:7.3f}")
