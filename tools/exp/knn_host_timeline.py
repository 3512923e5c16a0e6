"""EXPERIMENT: host-side duration of every C-ABI call in one steady-state iteration of the KNN attack on SSG (B=64, N=2048):
a call that takes long on the host is one that blocks (or a slow API inside the entry point)."""
import importlib, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import unit_cloud
M = importlib.import_module
dev = torch.device("cuda:0")
seeded_state_dict = M("3dpointcloudattack_amd.seeding").seeded_state_dict
net = M("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg(num_classes=40)
net.load_state_dict(seeded_state_dict(net, 0)); net = net.to(dev).eval()
rng = np.random.default_rng(0)
B, N = 64, 2048
pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
with torch.no_grad():
    lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
ka = M("3dpointcloudattack_amd.attack.KNN.KNN_attack")
adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
lib = M("3dpointcloudattack_amd._lib")
data = torch.cat([pcs, torch.nn.functional.normalize(pcs, dim=2)], dim=2)
log = []
orig = lib.call
evs = []
def timed(name, *a):
    t0 = time.perf_counter()
    r = orig(name, *a)
    log.append((t0, time.perf_counter() - t0, name))
    if name == "pc3d_adam_clip_step_f32":       # how far ahead of the GPU is the host? an event per iteration
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append((time.perf_counter(), e))
    return r
lib.call = timed
M("3dpointcloudattack_amd.ops")._lib.call = timed
atk = ka.CWKNN(net, None, None, None, None, None, adv.UntargetedLogitsAdvLoss(kappa=15.), du.ChamferkNNDist(), cu.ProjectInnerClipLinf(budget=0.18), attack_lr=1e-2, num_iter=40)
atk.attack(data, lab)
torch.cuda.synchronize()
h0, e0 = evs[0]
print("iteration: host submit time / GPU completion time since the first update (ms), host lead (ms)")
for i in (1, 5, 10, 20, 30, 39):
    h, e = evs[i]
    print("  it %2d  host %7.2f  gpu %7.2f  lead %6.2f" % (i, (h - h0) * 1e3, e0.elapsed_time(e), e0.elapsed_time(e) - (h - h0) * 1e3))
# one iteration = between consecutive adam calls near the end
idx = [i for i, e in enumerate(log) if e[2] == "pc3d_adam_clip_step_f32"]
a, b = idx[-6], idx[-5]
t0 = log[a][0] + log[a][1]
print("host time of the iteration: %.1f us, %d calls" % ((log[b][0] + log[b][1] - t0) * 1e6, b - a))
prev = t0
for t, d, n in log[a + 1:b + 1]:
    print("%8.1f  +%6.1f (gap before %6.1f)  %s" % ((t - t0) * 1e6, d * 1e6, (t - prev) * 1e6, n))
    prev = t + d
