"""Set-abstraction chain backward at SSG's two levels (B=64): one launch + points pass (ops.SA_CHAIN_BWD) against the
four-launch form; us per backward, HIP events around 10 repetitions."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_sa_chain_gpu import _case
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for name, (B, N, S, ns, C1, C2, C3) in (("SA1", (64, 2048, 512, 32, 64, 64, 128)), ("SA2", (64, 512, 128, 64, 128, 128, 256))):
    P, Bc, idx, layers, w = _case(ops, dev, B, N, S, ns, C1, C2, C3, seed=1)
    rev = ops.group_reverse(idx, N)
    for fused, sparse in ((True, True), (True, False), (False, False)):
        ops.SA_CHAIN_BWD, ops.SA_BWD_SPARSE = fused, sparse
        ts = []
        for it in range(6):
            p, bc = P.clone().requires_grad_(), Bc.clone().requires_grad_()
            out = ops.grouped_mlp_max(p, bc, idx, layers, rev=rev)
            loss = (out * w).sum()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            loss.backward()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(name, ("sparse fused" if sparse else "fused") if fused else "four-launch", "backward us (incl. the loss's two ATen launches):", round(sorted(ts)[len(ts) // 2], 1), flush=True)
ops.SA_CHAIN_BWD = ops.SA_BWD_SPARSE = True
