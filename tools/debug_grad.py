import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.nn.functional as F
from helpers import hip_pointnet, oracle_pointnet
from oracle import ref_torch as ort
dev = torch.device("cuda:0")
fx = np.load(os.path.join(ROOT, "tests/golden/cw.npz"))
model, _ = hip_pointnet(0, dev); omodel, _ = oracle_pointnet(0)
nm = "l2_untarget"
x1 = fx[f"{nm}_traj"][1][None]   # adv at iteration 1 (golden == hip to 1e-8)
tgt = torch.from_numpy(fx[f"{nm}_target"])
def grad_of(m, x, dev_):
    x = torch.from_numpy(x).to(dev_).requires_grad_()
    logp = m(x)[0]
    loss = ort.UntargetedLogitsAdvLoss(5.)(logp.cpu(), tgt) if dev_ == "cpu" else None
    if loss is None:
        adv = importlib.import_module("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
        loss = adv.UntargetedLogitsAdvLoss(5.)(logp, tgt.to(dev_))
    loss.backward()
    return x.grad.cpu().numpy()[0], logp.detach().cpu().numpy()
gh, lh = grad_of(model, x1, dev)
go, lo = grad_of(omodel, x1, "cpu")
d = np.abs(gh - go)
print("logp diff", np.abs(lh - lo).max(), "grad max", np.abs(go).max(), "diff max", d.max())
bad = np.where(d.max(0) > 1e-6 * np.abs(go).max())[0]
print("bad points", bad, d.max(0)[bad], "oracle grad there", go[:, bad].T, "hip", gh[:, bad].T)
# which tower? compare STN-only and trunk-only critical sets
ops = importlib.import_module("3dpointcloudattack_amd.ops")
xt = torch.from_numpy(x1).to(dev)
stn_tower, stn_head, iden = model.feat.stn.folded()
pooled, idx = ops.pointmlp3_max_fwd_raw(xt, stn_tower, True)
with torch.no_grad():
    s = omodel.feat.stn
    xc = torch.from_numpy(x1)
    h = F.relu(s.bn3(s.conv3(F.relu(s.bn2(s.conv2(F.relu(s.bn1(s.conv1(xc)))))))))
    ov, oi = h.max(2)
print("stn argmax mismatches", (idx.cpu().long() != oi).sum().item(), "of which pooled>0:", ((idx.cpu().long() != oi) & (ov > 0)).sum().item())
mm = torch.where((idx.cpu().long() != oi) & (ov > 0))
for b, c in zip(*mm):
    print(" ch", int(c), "hip idx", int(idx[b, c]), "torch idx", int(oi[b, c]), "vals", float(h[b, c, idx[b, c].long().cpu()]), float(ov[b, c]))
trans = model.feat.stn(xt)
xtt = torch.bmm(xt.transpose(2, 1), trans).transpose(2, 1)
pooled2, idx2 = ops.pointmlp3_max_fwd_raw(xtt, model.feat.folded(), False)
with torch.no_grad():
    f = omodel.feat
    tr = f.stn(xc); xo = torch.bmm(xc.transpose(2, 1), tr).transpose(2, 1)
    h = f.bn3(f.conv3(F.relu(f.bn2(f.conv2(F.relu(f.bn1(f.conv1(xo))))))))
    ov2, oi2 = h.max(2)
mm = torch.where(idx2.cpu().long() != oi2)
print("trunk argmax mismatches", len(mm[0]))
for b, c in zip(*mm):
    print(" ch", int(c), "hip idx", int(idx2[b, c]), "torch idx", int(oi2[b, c]), "vals", float(h[b, c, idx2[b, c].long().cpu()]), float(ov2[b, c]))
