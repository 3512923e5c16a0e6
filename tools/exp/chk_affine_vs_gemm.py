import importlib, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((4, 3, 1024)).astype(np.float32)).to(dev)
w = torch.from_numpy(rng.standard_normal((128, 3)).astype(np.float32)).to(dev)
b = torch.from_numpy(rng.standard_normal(128).astype(np.float32)).to(dev)
a = ops.affine3(x.permute(0, 2, 1), w, b)
g = ops.linear_act(x.transpose(2, 1).contiguous(), w, b)
print("equal:", torch.equal(a, g), "max abs diff", float((a - g).abs().max()), "n diff", int((a != g).sum()), "of", a.numel())
