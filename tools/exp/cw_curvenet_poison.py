#!/usr/bin/env python3
"""GPU box: does CW on CurveNet (B=32, N=4096, 3 iterations) depend on what freed device memory holds? Runs the loop after
filling the allocator's free blocks with 0 / NaN / 1e30 / -1e30 and compares the iterates bit-wise (graphed and eager)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_configs_gpu as tc
dev = torch.device("cuda:0")
cwm, adv, dist, clip = tc._cw_mods()


def poison(val):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = []
    for mb in (2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1):
        for _ in range(6):
            try:
                blocks.append(torch.full((mb * 262144,), val, dtype=torch.float32, device=dev))
            except RuntimeError:
                break
    small = [torch.full((n,), val, dtype=torch.float32, device=dev) for n in (64, 256, 1024, 4096, 16384, 65536) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks, small


B, N, G = 32, 4096, 256
pcs = tc._clouds(B, N, 1238)
seeds = [1000 + i for i in range(B)]
VALS = [float(v) for v in os.environ.get("POISON_VALS", "0,nan,1e30,-1e30,0").split(",")]
for graph in ((True,) if os.environ.get("GRAPH_ONLY") else (True, False)):
    model = tc._hip_curvenet(dev)
    with torch.no_grad():
        labels = model(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
    ref = None
    for val in VALS:
        if not os.environ.get("NO_POISON"):
            poison(val)
        atk = cwm.CW(model, model, adv_func=adv.UntargetedLogitsAdvLoss(0.), clip_func=clip.ClipPointsLinf(0.18),
                     dist_func=dist.ChamferDist(), attack_lr=1e-2, binary_step=1, num_iter=3, graph=graph, sample_seeds=seeds, global_batch=G)
        st = tc._cw_state_after(atk, pcs, labels, 3)
        a1, a3 = st["adv1"].clone(), st["adv"].detach().clone()
        if ref is None:
            ref = (a1, a3)
        print(f"graph={graph} poison={val}: adv1 equal {bool(torch.equal(a1, ref[0]))} (max dev {float((a1 - ref[0]).abs().max()):.3e}), "
              f"adv3 equal {bool(torch.equal(a3, ref[1]))} (max dev {float((a3 - ref[1]).abs().max()):.3e}), finite {bool(torch.isfinite(a3).all())}", flush=True)
        del atk, st
