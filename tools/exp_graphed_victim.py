"""Experiment: forward+backward of the victims through torch.cuda.make_graphed_callables (one hipGraph per direction)
vs eager launches. Prints ms per forward+backward and the max abs difference of the input gradient."""
import importlib, sys, os, json, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import unit_cloud
seeded_state_dict = importlib.import_module("3dpointcloudattack_amd.seeding").seeded_state_dict
M = importlib.import_module
dev = torch.device("cuda:0")
def mk(modname, cls, seed=0, **kw):
    m = getattr(M(f"3dpointcloudattack_amd.model.{modname}"), cls)(**kw)
    m.load_state_dict(seeded_state_dict(m, seed))
    for p in m.parameters(): p.requires_grad_(False)
    return m.to(dev).eval()
def timeit(fn, n=8, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
rng = np.random.default_rng(0)
def clouds(B, N): return torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).transpose(1, 2).contiguous().to(dev)
cfg = {"curvenet": ("curvenet", "CurveNet", dict(num_classes=40), 32, 4096), "dgcnn": ("dgcnn", "DGCNN", None, 32, 1024),
       "ssg": ("pointnet2_SSG", "PointNet_Ssg", dict(num_classes=40), 64, 2048), "pointnet": ("pointnet", "PointNetCls", dict(k=40), 32, 1024)}
res = {}
for nm in sys.argv[1:] or ["curvenet", "dgcnn"]:
    mod, cls, kw, B, N = cfg[nm]
    model = mk(mod, cls, 0, args=types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40) if nm == "dgcnn" else mk(mod, cls, 0, **kw)
    x = clouds(B, N)
    f = lambda inp: model(inp)[0]
    def step(fn, xin):
        xa = xin.clone().requires_grad_()
        fn(xa).logsumexp(1).sum().backward()
        return xa.grad
    eager = timeit(lambda: step(f, x))
    g_e = step(f, x).clone()
    try:
        fg = torch.cuda.make_graphed_callables(f, (x.clone().requires_grad_(),))
        graphed = timeit(lambda: step(fg, x))
        g_g = step(fg, x).clone()
        res[nm] = {"eager_ms": round(eager, 3), "graphed_ms": round(graphed, 3), "grad_maxabs_diff": float((g_e - g_g).abs().max()),
                   "grad_maxabs": float(g_e.abs().max())}
        # forward + loss + backward in ONE graph (what a whole-iteration capture does)
        sx = x.clone().requires_grad_()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                torch.autograd.grad(f(sx).logsumexp(1).sum(), sx)
        torch.cuda.current_stream().wait_stream(side)
        one = torch.cuda.CUDAGraph()
        with torch.cuda.graph(one):
            (sg,) = torch.autograd.grad(f(sx).logsumexp(1).sum(), sx)
        res[nm]["one_graph_ms"] = round(timeit(lambda: one.replay()), 3)
        res[nm]["one_graph_grad_diff"] = float((sg - g_e).abs().max())
    except Exception as e:
        res[nm] = {"eager_ms": round(eager, 3), "error": repr(e)[:400]}
    print(nm, res[nm], flush=True)
print(json.dumps(res))
