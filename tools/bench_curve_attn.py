"""pc3d_curve_attn_f32 forward / backward at the CIC shapes of cfg5 (B=32, N=1024, C=16/32, cn=100, cl=5): us."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N, C, cn, cl in ((32, 1024, 16, 100, 5), (32, 1024, 32, 100, 5), (32, 1024, 64, 100, 5)):
    R = cn + cl
    x = torch.randn(B, N, C, device=dev, requires_grad=True)
    K = (0.5 * torch.randn(B, C, R, device=dev)).requires_grad_()
    V = torch.randn(B, R, C, device=dev, requires_grad=True)
    up = torch.randn(B, N, C, device=dev)
    def fwd():
        return ops.curve_attn(x, K, V, cn, 0.2)
    def both():
        x.grad = K.grad = V.grad = None
        fwd().backward(up)
    res = {}
    for name, fn in (("fwd", fwd), ("fwd+bwd", both)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps({"B": B, "N": N, "C": C, "R": R, **res}), flush=True)
