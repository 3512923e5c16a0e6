"""EXPERIMENT: which switch moves the second loss value of the cfg3 golden case (tests/test_configs_gpu.py)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_configs_gpu as T
from test_oracle_golden import GEO_CASES, _geo_cfg
M = importlib.import_module
dev = torch.device("cuda:0")
ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
dg = M("3dpointcloudattack_amd.model.dgcnn")
fx = np.load(os.path.join(ROOT, "tests", "golden", "geoa3_dgcnn.npz"))
net, sha = T._hip_dgcnn(dev)
nm = "ce_cd_hd_curv"
pc, label = torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_label"])
def run(direct_in, direct_terms):
    dg.FIRST_LAYER_DIRECT = direct_in
    cfg = _geo_cfg(host_rng=True, **GEO_CASES[nm])
    cfg.direct_terms = direct_terms
    torch.manual_seed(77); np.random.seed(77)
    best, tgt, mask, steps, losses = ga.geoA3_attack(net, None, None, None, None, None, pc, label, cfg, 0, 1)
    return np.array(losses)
def poison(val):
    """Fill ~6 GB of the caching allocator's free blocks (many sizes) with `val`: a kernel that reads memory nobody wrote
    then sees it instead of the zeros of a fresh process."""
    ts = [torch.full((n,), val, device=dev) for n in [1 << k for k in range(8, 28)] for _ in range(3)]
    ts += [torch.full((n,), val, device=dev).to(torch.int32) for n in [1 << k for k in range(8, 24)]]
    del ts
for val in (None, float("nan"), 1e30, 3.0):
    for di in (True, False):
        if val is not None:
            poison(val)
        L = run(di, True)
        print("poison", val, "first-layer-direct", di, "L[:4] =", L[:4, 0] if L.ndim > 1 else L[:4], flush=True)
