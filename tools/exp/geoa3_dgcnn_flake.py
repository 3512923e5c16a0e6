"""EXPERIMENT: run-to-run spread of the GeoA3-on-DGCNN golden cases (tests/test_configs_gpu.py) within one process: the
only nondeterminism is the order of float atomics in the backward; a 1e-7 difference can re-wire DGCNN's feature-space kNN
graph of a later iterate."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_configs_gpu as t
dev = torch.device("cuda:0")
ga = t.M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
fx = np.load(os.path.join(t.GOLDEN, "geoa3_dgcnn.npz"))
net, sha = t._hip_dgcnn(dev)
for nm, rep in [(n, r) for n in ("margin_l2", "ce_cd_hd_curv") for r in range(10)]:
    R = fx[f"{nm}_losses"]
    cfg = t._geo_cfg(host_rng=True, **t.GEO_CASES[nm])
    pc, label = torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_label"])
    torch.manual_seed(77); np.random.seed(77)
    best, tgt, mask, steps, losses = ga.geoA3_attack(net, None, None, None, None, None, pc, label, cfg, 0, 1)
    L = np.array(losses)
    rel = np.abs(L - R) / np.abs(R)
    print(nm, rep, "L[:3]", L[:3, 0].tolist(), "first dev>1e-4 at iter", int(np.argmax(rel[:, 0] > 1e-4)) if (rel > 1e-4).any() else -1, "max rel", float(rel.max()), flush=True)
