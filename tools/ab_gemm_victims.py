"""A/B in ONE process on one box: victims' forward / forward+backward with the hand-written fp32-MFMA point-wise kernel
(pc3d_gemm_nt_f32) against the same code with ops.gemm_nt monkey-patched to the library path (torch.addmm -> hipBLASLt,
+ separate activation / mask passes). The patch lives here only; the product always runs its own kernel."""
import importlib, sys, os, json, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.nn.functional as F
from helpers import unit_cloud
seeded_state_dict = importlib.import_module("3dpointcloudattack_amd.seeding").seeded_state_dict
ops = importlib.import_module("3dpointcloudattack_amd.ops")
M = importlib.import_module
dev = torch.device("cuda:0")
own = ops.gemm_nt


def lib_gemm(x2d, w, bias=None, act=None, slope=0.0, gate=None, gate_slope=0.0, out=None):
    if gate is not None:
        x2d = torch.where(gate > 0, x2d, gate_slope * x2d)
    y = torch.addmm(bias, x2d, w.t()) if bias is not None else x2d @ w.t()
    if act == "relu":
        y = torch.relu_(y)
    elif act == "leaky":
        y = F.leaky_relu_(y, slope)
    if out is not None:
        out.copy_(y)
        return out
    return y


def mk(modname, cls, seed=0, **kw):
    m = getattr(M(f"3dpointcloudattack_amd.model.{modname}"), cls)(**kw)
    m.load_state_dict(seeded_state_dict(m, seed)); return m.to(dev).eval()
def timeit(fn, n=8, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fwdbwd(model, x):
    xa = x.clone().requires_grad_()
    out = model(xa); out = out[0] if isinstance(out, tuple) else out
    out.logsumexp(1).sum().backward()
rng = np.random.default_rng(0)
def clouds(B, N): return torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).transpose(1, 2).contiguous().to(dev)
cfg = {"ssg": ("pointnet2_SSG", "PointNet_Ssg", dict(num_classes=40), 64, 2048),
       "msg": ("pointnet2_MSG", "PointNet_Msg", dict(num_class=40, normal_channel=False), 32, 1024),
       "dgcnn": ("dgcnn", "DGCNN", dict(args=types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40), 32, 1024),
       "curvenet": ("curvenet", "CurveNet", dict(num_classes=40), 32, 4096)}
for nm in (sys.argv[1:] or list(cfg)):
    mod, cls, kw, B, N = cfg[nm]
    model = mk(mod, cls, 0, **kw)
    x = clouds(B, N)
    row = {"victim": nm, "B": B, "N": N}
    for tag, fn in (("own", own), ("lib", lib_gemm), ("own2", own)):
        ops.gemm_nt = fn
        torch.manual_seed(0)
        with torch.no_grad():
            row[f"fwd_ms_{tag}"] = round(timeit(lambda: model(x)), 3)
        row[f"fwd_bwd_ms_{tag}"] = round(timeit(lambda: fwdbwd(model, x)), 3)
    ops.gemm_nt = own
    print(json.dumps(row), flush=True)
