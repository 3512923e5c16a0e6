import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import hip_pointnet, oracle_pointnet
from oracle import ref_torch as ort
m = importlib.import_module
cwm, adv, dist, clip = (m("3dpointcloudattack_amd.attack.CW.CW_attack"), m("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"),
            m("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils"), m("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils"))
dev = torch.device("cuda:0")
fx = np.load(os.path.join(ROOT, "tests/golden/cw.npz"))
model, _ = hip_pointnet(0, dev); tm, _ = hip_pointnet(1, dev); omodel, _ = oracle_pointnet(0)
nm = sys.argv[1] if len(sys.argv) > 1 else "chamfer_untarget"
steps, iters, kappa = fx[f"{nm}_cfg"]
traj = []
class Rec(torch.nn.Module):
    def __init__(self, inner):
        super().__init__(); self.inner = inner
    def forward(self, a, o, w=None, batch_avg=True):
        traj.append(a.detach().cpu().numpy()[0].copy()); return self.inner(a, o, w, batch_avg)
atk = cwm.CW(model, tm, adv_func=(adv.UntargetedLogitsAdvLoss(kappa) if "untarget" in nm else adv.LogitsAdvLoss(kappa)), clip_func=clip.ClipPointsLinf(0.18), dist_func=Rec(dist.ChamferDist() if nm.startswith("chamfer") else dist.L2Dist()), binary_step=int(steps), num_iter=int(iters), fused=False, attack_method=("untarget" if "untarget" in nm else "target"))
torch.manual_seed(1000); np.random.seed(1000)
bd, ba, sn = atk.attack(torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_target"]))
otraj = []
torch.manual_seed(1000)
obd, oba, osn, _ = ort.cw_attack(omodel, torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_target"]), (ort.UntargetedLogitsAdvLoss(kappa) if "untarget" in nm else ort.LogitsAdvLoss(kappa)), (ort.ChannelFirst(ort.ChamferDist(dtype=torch.float64)) if nm.startswith("chamfer") else ort.L2Dist()), ort.ClipPointsLinf(0.18), binary_step=int(steps), num_iter=int(iters), record=lambda s, i, a: otraj.append(a[0].copy()), attack_method=("untarget" if "untarget" in nm else "target"))
traj, otraj, g = np.stack(traj), np.stack(otraj), fx[f"{nm}_traj"]
for i in range(len(traj)):
    d = np.abs(traj[i]-otraj[i]); d2 = np.abs(g[i]-otraj[i])
    print(i, "hip-vs-oracle max %.2e n>1e-4 %d p99 %.1e| golden-vs-oracle max %.2e n %d" % (d.max(), (d>1e-4).sum(), np.quantile(d, 0.99), d2.max(), (d2>1e-4).sum()))
print(bd, obd, fx[f"{nm}_bestdist"])
