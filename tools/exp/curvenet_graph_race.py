#!/usr/bin/env python3
"""GPU box: which tensor of a replayed CurveNet forward graph first differs between replays (same input)? Every block's
output and every geometry tensor (FPS picks, ball-query groups, kNN graphs) is kept alive through the capture, so each replay
rewrites it in place and it can be compared afterwards. PC3D_FPS_THREADS=64 makes the glitches frequent."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_configs_gpu as tc
dev = torch.device("cuda:0")
B, N = 32, 4096
model = tc._hip_curvenet(dev)
if os.environ.get("INLINE_GEOMETRY"):
    model.geometry_stream = False
x = tc._clouds(B, N, 1238).transpose(1, 2).contiguous().to(dev)
stash = {}
names = ["lpfa", "cic11", "cic12", "cic21", "cic22", "cic31", "cic32", "cic41", "cic42"]
for nm in names:
    getattr(model, nm).register_forward_hook((lambda n: lambda mod, inp, out: stash.__setitem__(n, out[1] if isinstance(out, tuple) else out))(nm))
ops = importlib.import_module("3dpointcloudattack_amd.ops")
blk = model.cic11
for nm, mod in (("c11_maxpool", blk.maxpool), ("c11_curves", blk.curvegrouping), ("c11_agg", blk.curveaggregation), ("c11_lpfa", blk.lpfa),
                ("c11_walk", blk.curvegrouping.walk)):
    mod.register_forward_hook((lambda n: lambda m_, inp, out: stash.__setitem__(n, out[1] if isinstance(out, tuple) else out))(nm))
_topk, _att = ops.topk_desc, ops.att_scale
cnt = {"t": 0, "a": 0}
def topk(att, k):
    r = _topk(att, k)
    stash[f"topk{cnt['t'] % 8}"] = r
    cnt["t"] += 1
    return r
def att_scale(x, w):
    r = _att(x, w)
    stash[f"att{cnt['a'] % 8}_xs"], stash[f"att{cnt['a'] % 8}_att"] = r
    cnt["a"] += 1
    return r
ops.topk_desc, ops.att_scale = topk, att_scale
orig_geo = model._geometry
def geo(pos, with_grad=False):
    lv = orig_geo(pos, with_grad)
    for i, (pool, graph, ev_pool, ev) in enumerate(lv):
        if pool is not None:
            stash[f"L{i}_fps"], stash[f"L{i}_ball"] = pool
        stash[f"L{i}_knn"] = graph[0]
    return lv
model._geometry = geo
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(3):
        model(x)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    out = model(x)[0]
keys = sorted(stash.keys())
print("stashed:", keys)
def snap():
    torch.cuda.synchronize()
    d = {k: stash[k].clone() for k in keys}
    d["logits"] = out.clone()
    return d
g.replay(); ref = snap()
order = [k for k in keys if k.startswith("L")] + ["lpfa", "c11_maxpool", "att0_att", "att0_xs", "topk0", "c11_walk", "c11_curves", "c11_agg", "c11_lpfa"] + names[1:] + ["logits"]
order = [k for k in order if k in ref]
bad = 0
for it in range(int(os.environ.get("REPLAYS", "300"))):
    g.replay()
    cur = snap()
    diffs = [k for k in order if not torch.equal(cur[k], ref[k])]
    if diffs:
        bad += 1
        k = diffs[0]
        d = (cur[k].float() - ref[k].float()).abs()
        idx = (d > 0).nonzero()
        print(f"replay {it}: differing tensors {diffs}; first {k}: {len(idx)} elements, samples {sorted(set(idx[:, 0].tolist()))[:6]}, "
              f"first index {idx[0].tolist()} cur {cur[k][tuple(idx[0])].item()} ref {ref[k][tuple(idx[0])].item()}", flush=True)
print("glitched replays:", bad)
