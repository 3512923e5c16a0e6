#!/usr/bin/env python3
"""GPU box: deviations of the mirror's GeoA3 loss curves from (a) the real reference's curve and (b) the oracle's
intended-semantics curve stored in tests/golden/config_sizes.npz — per iteration, for every case and size. The tolerances of
tests/test_config_sizes_gpu.py and tests/test_configs_gpu.py are set from this table (the deterministic build gives the same
numbers on every run and box)."""
import importlib, os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_configs_gpu as tc
from oracle import ref_torch as ort
M = importlib.import_module
dev = torch.device("cuda:0")
ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
G = os.path.join(ROOT, "tests", "golden")
cs = np.load(os.path.join(G, "config_sizes.npz"))
g256 = np.load(os.path.join(G, "geoa3_dgcnn.npz"))
cw = np.load(os.path.join(G, "cw_curvenet.npz"))
dg, _ = tc._hip_dgcnn(dev)
cn = tc._hip_curvenet(dev, cw["conv2_bias"])


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-3)


def run(net, pc, label, nm, N, seed):
    cfg = tc._geo_cfg(host_rng=True, npoint=N, **tc.GEO_CASES[nm])
    torch.manual_seed(seed); np.random.seed(seed)
    best, tgt, mask, steps, losses = ga.geoA3_attack(net, None, None, None, None, None, torch.from_numpy(pc), torch.from_numpy(label), cfg, 0, 1)
    return best.cpu().numpy(), np.asarray(mask), np.array(losses)[:, 0]


for tag, net, fx, prefix, N, seed in (("dgcnn n256", dg, g256, "", 256, 77), ("dgcnn n1024", dg, cs, "dgcnn_n1024_", 1024, 78),
                                      ("curvenet n1024", cn, cs, "cngeo_n1024_", 1024, 79)):
    for nm in ("ce_cd_hd_curv", "margin_l2"):
        k = prefix + nm
        best, mask, L = run(net, fx[f"{k}_pc"], fx[f"{k}_label"], nm, N, seed)
        R = fx[f"{k}_losses"][:, 0]
        ok = f"dgcnn_n256_{nm}" if prefix == "" else k
        OL = cs[f"{ok}_olosses"][:, 0]
        OB = cs[f"{ok}_obest"]
        print(f"== {tag} {nm}: mask {mask} ref {fx[f'{k}_mask']} oracle {cs[f'{ok}_omask']}")
        print("   rel dev vs reference :", " ".join(f"{v:.1e}" for v in rel(L, R)))
        print("   rel dev vs oracle    :", " ".join(f"{v:.1e}" for v in rel(L, OL)))
        print("   oracle vs reference  :", " ".join(f"{v:.1e}" for v in rel(OL, R)))
        d1, d2 = np.abs(best - fx[f"{k}_best"]), np.abs(best - OB)
        print(f"   best: vs ref median {np.median(d1):.2e} q99 {np.quantile(d1, .99):.2e} max {d1.max():.2e} | vs oracle median {np.median(d2):.2e} q99 {np.quantile(d2, .99):.2e} max {d2.max():.2e}")
# CurveNet N = 4096 logits / gradient
cnm = M("3dpointcloudattack_amd.model.curvenet").CurveNet(num_classes=40)
cnm.load_state_dict(ort.seeded_state_dict(cnm, 9, gain=1.0))
cnm = cnm.eval().to(dev)
x = torch.from_numpy(cs["curvenet_n4096_x"]).to(dev).requires_grad_()
lg = cnm(x)[0]
(lg * torch.from_numpy(cs["curvenet_n4096_w"]).to(dev)).sum().backward()
ref = cs["curvenet_n4096_logits"]
print("curvenet n4096 logits max abs dev", float(np.abs(lg.detach().cpu().numpy() - ref).max()), "of", float(np.abs(ref).max()),
      "grad rel", float(np.linalg.norm(x.grad.cpu().numpy() - cs["curvenet_n4096_gx"]) / np.linalg.norm(cs["curvenet_n4096_gx"])))
