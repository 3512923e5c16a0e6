#!/usr/bin/env python3
"""GPU box: where does the graphed CurveNet loop stop being run == run when FPS timing changes (PC3D_FPS_THREADS=64)?
Repeats the 3-iteration CW loop on one victim; per iteration records checksums of the victim's logits, of the victim's input
gradient and of the iterate; prints the first quantity that differs from the first run."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_configs_gpu as tc
dev = torch.device("cuda:0")
cwm, adv, dist, clip = tc._cw_mods()
graphed = importlib.import_module("3dpointcloudattack_amd.graphed")
B, N, G = 32, 4096, 256
pcs = tc._clouds(B, N, 1238)
seeds = [1000 + i for i in range(B)]
model = tc._hip_curvenet(dev)
if os.environ.get("INLINE_GEOMETRY"):
    model.geometry_stream = False
with torch.no_grad():
    labels = model(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
gv = graphed.wrap(model)
log = []
orig_forward = gv.forward
def fwd(x):
    out = orig_forward(x)
    lg = out[0]
    rec = {"logits": lg.detach().clone(), "x": x.detach().clone()}
    if lg.requires_grad:
        lg.register_hook(lambda g: rec.__setitem__("g_logits", g.detach().clone()))
        x.register_hook(lambda g: rec.__setitem__("gx", g.detach().clone())) if x.requires_grad else None
    log.append(rec)
    return out
gv.forward = fwd
ref = None
for run in range(int(os.environ.get("RUNS", "16"))):
    log.clear()
    atk = cwm.CW(model, model, adv_func=adv.UntargetedLogitsAdvLoss(0.), clip_func=clip.ClipPointsLinf(0.18),
                 dist_func=dist.ChamferDist(), attack_lr=1e-2, binary_step=1, num_iter=3, graph=True, sample_seeds=seeds, global_batch=G)
    if os.environ.get("NO_DIST_STREAM"):
        atk.dist_stream = False
    st = tc._cw_state_after(atk, pcs, labels, 3)
    torch.cuda.synchronize()
    cur = [{k: v.cpu() for k, v in r.items()} for r in log]
    if ref is None:
        ref = cur
        print("forward calls per run:", len(cur), [sorted(r.keys()) for r in cur])
        continue
    msg = "same"
    for i, (a, b) in enumerate(zip(cur, ref)):
        for k in ("x", "logits", "g_logits", "gx"):
            if k in a and k in b and not torch.equal(a[k], b[k]):
                d = (a[k] - b[k]).abs()
                msg = f"first difference: forward call {i}, {k}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}, samples {sorted(set((d > 0).nonzero()[:, 0].tolist()))[:8]}"
                break
        if msg != "same":
            break
    print(f"run {run}: {msg}  | stats {gv.stats}", flush=True)
